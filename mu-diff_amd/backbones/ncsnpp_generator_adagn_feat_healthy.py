"""Two-condition twins of the generators (reference backbones/ncsnpp_generator_adagn_feat_healthy.py): the same
networks with ONE condition less - G1 concatenates x + 2 condition feature maps (3*nf channels, reference :182-184,
:329), G2 fuses the single pair c12 (2*nf channels, :137-139, :301-311).  `forward(x, cond1, cond2, time_cond, z
[, pseudo_target])`.  Nothing in the reference imports this file (its classes collide with the three-condition ones in
the model registry, so both cannot even be imported in one process there); here they share every kernel and all host
logic with ncsnpp_generator_adagn_feat.py and are not registered by name."""
from .ncsnpp_generator_adagn_feat import _G1, _G2


class NCSNpp(_G1):
    N_COND = 2

    def forward(self, x, cond1, cond2, time_cond, z):
        return self._forward(x, (cond1, cond2), time_cond, z)


class NCSNpp_adaptive(_G2):
    N_COND = 2

    def forward(self, x, cond1, cond2, time_cond, z, pseudo_target):
        return self._forward(x, (cond1, cond2), time_cond, z, pseudo_target)
