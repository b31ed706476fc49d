"""Builds experiment variants of libmudiff_hip.so (conv_mfma.hip compiled with -D knobs) for in-process A/B runs
(scripts/ab_conv.py).  Output: mu-diff_amd/mudiff_hip/variants/lib_<name>.so (git-ignored, travels with gpurun).
    python scripts/build_variants.py name1:-DCM_STAGGER=1 name2:"-DCM_PRIO=1 -DCM_STAGGER=1" ..."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'mu-diff_amd', 'csrc')
OUT = os.path.join(ROOT, 'mu-diff_amd', 'mudiff_hip', 'variants')
os.makedirs(OUT, exist_ok=True)
subprocess.check_call(['make', '-C', CSRC, '-j8'], stdout=subprocess.DEVNULL)
objs = [os.path.join(CSRC, o) for o in ('elementwise.o', 'groupnorm.o', 'dense.o', 'conv_direct.o', 'fir.o', 'attention.o', 'api.o')]


def build(spec):
    name, _, flags = spec.partition(':')
    obj = f'/tmp/_conv_{name}.o'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-fPIC', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-function', '-Wno-pass-failed',
                           *flags.split(), '-c', os.path.join(CSRC, 'conv_mfma.hip'), '-o', obj], stderr=subprocess.DEVNULL)
    lib = os.path.join(OUT, f'lib_{name}.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', *objs, obj, '-o', lib])
    return lib


with ThreadPoolExecutor(4) as ex:
    for lib in ex.map(build, sys.argv[1:]):
        print('built', lib)
