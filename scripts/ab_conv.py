"""In-process A/B of library variants on the 3x3 conv layer shapes of config 2 (interleaved rounds, one device, random data:
cdna_hip_programming.md section 5.4 rules 24/25).
    python scripts/ab_conv.py B rounds lib_a.so lib_b.so ...        (first library = baseline)
Prints per shape the median us of every variant and its ratio to the baseline, then the time-weighted total."""
import ctypes as C, math, os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mu-diff_amd')
import numpy as np, torch
import mudiff_hip
from mudiff_hip import ConvArgs

B, rounds = int(sys.argv[1]), int(sys.argv[2])
paths = sys.argv[3:]
libs = []
for p in paths:
    lib = C.CDLL(os.path.abspath(p))
    for name, (res, args) in mudiff_hip._SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    libs.append(lib)
names = [os.path.basename(p).replace('lib_', '').replace('.so', '') for p in paths]
dev = 'cuda:0'
# (H, Cin, Cout, residual, launches per G1+G2 pass) - the 3x3 MFMA launches of profiles/r01_e_layer_times_b16.txt
shapes = [(256, 192, 384, 0, 1), (256, 320, 64, 0, 2), (256, 128, 128, 1, 2), (256, 256, 64, 0, 2), (256, 128, 128, 0, 2), (64, 256, 256, 1, 14),
          (128, 256, 256, 1, 2), (128, 256, 256, 0, 2), (64, 512, 256, 0, 4), (256, 192, 64, 0, 2), (256, 64, 64, 0, 12), (256, 64, 64, 1, 10),
          (128, 384, 128, 0, 2), (128, 128, 128, 1, 10), (256, 128, 64, 0, 2), (128, 192, 128, 0, 2), (64, 384, 256, 0, 2), (128, 256, 128, 0, 2)]
if len(os.environ.get('AB_SHAPES', '')):
    keep = [int(i) for i in os.environ['AB_SHAPES'].split(',')]
    shapes = [shapes[i] for i in keep]
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())
tot = np.zeros(len(libs))
for H, Cin, Cout, res, cnt in shapes:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9)
    if os.environ.get('AB_ZERO') == '1':      # DVFS probe: same instruction stream, no switching energy in the operands
        x.zero_(); w.zero_()
    sc, sh = torch.rand(B, Cin, device=dev) + 0.5, torch.randn(B, Cin, device=dev)
    if os.environ.get('AB_ZERO') == '1':
        sh.zero_()                                # silu(sc * 0 + 0) = 0: the A operand is all zeros too
    r = torch.randn(B, H, H, Cout, device=dev) if res else None
    out = torch.empty(B, H, H, Cout, device=dev)
    stats = torch.zeros(B, Cout, 2, device=dev, dtype=torch.float64)
    b2 = torch.randn(B, Cout, device=dev)
    packed, argsl = [], []
    for lib in libs:
        nb = lib.mud_packed_weight_bytes(3, Cin, Cout)
        pw = torch.empty(nb, device=dev, dtype=torch.uint8)
        assert lib.mud_pack_weights(P(w), 1, 9, Cin * 9, 0, 3, Cin, Cout, 1, P(pw), stream) == 0
        a = ConvArgs()
        a.x, a.B, a.H, a.W, a.Cin, a.ldx = P(x), B, H, H, Cin, Cin
        a.w, a.ks, a.stride, a.pad = P(pw), 3, 1, 1
        a.pro_scale, a.pro_shift, a.pro_ld, a.pro_mode = P(sc), P(sh), Cin, 2
        a.bias2, a.bias2_ld = P(b2), Cout
        if res:
            a.res, a.ldr = P(r), Cout
        a.out_scale, a.out, a.Cout, a.ldo = 0.7071, P(out), Cout, Cout
        a.stats, a.stats_ld = P(stats), Cout
        packed.append(pw); argsl.append(a)
    times = [[] for _ in libs]
    for rd in range(rounds + 1):
        for i, lib in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                assert lib.mud_conv2d_mfma(C.byref(argsl[i]), stream) == 0
            e1.record(); torch.cuda.synchronize()
            if rd:
                times[i].append(e0.elapsed_time(e1) / 5 * 1e3)
    med = np.array([np.median(t) for t in times])
    tot += med * cnt
    fl = 2.0 * B * H * H * Cout * Cin * 9
    print(f'{H:4d}^2 {Cin:4d}->{Cout:4d} res{res} x{cnt:2d}: ' + '  '.join(f'{n} {m:7.1f}us ({fl / m / 1e6:5.0f}TF, {m / med[0]:.3f})' for n, m in zip(names, med)), flush=True)
print('weighted total per G1+G2 pass: ' + '  '.join(f'{n} {t / 1e3:7.2f}ms ({t / tot[0]:.3f})' for n, t in zip(names, tot)))
