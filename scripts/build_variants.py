"""Builds experiment variants of libmudiff_hip.so (conv_mfma.hip compiled with extra flags) for in-process A/B runs
(scripts/ab_conv.py).  Output: mu-diff_amd/mudiff_hip/variants/lib_<name>.so (git-ignored, travels with gpurun).  A variant
reports its flags through mud_build_flags(), and mudiff_hip.load() refuses it unless MUDIFF_ALLOW_VARIANT=1.
    python scripts/build_variants.py name1:-DSOME_KNOB=1 name2:"-DA=1 -DB=2" ...
(The shipped source carries no experiment macros since round 3; a variant adds its own to a scratch copy or passes -D flags an
edited copy understands.  `name:@path/to/edited_conv_mfma.hip[:flags]` compiles that file instead of csrc/conv_mfma.hip.)"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'mu-diff_amd', 'csrc')
OUT = os.path.join(ROOT, 'mu-diff_amd', 'mudiff_hip', 'variants')
os.makedirs(OUT, exist_ok=True)
subprocess.check_call(['make', '-C', CSRC, '-j8'], stdout=subprocess.DEVNULL)
objs = [os.path.join(CSRC, o) for o in ('elementwise.o', 'groupnorm.o', 'dense.o', 'conv_direct.o', 'fir.o', 'attention.o')]


def build(spec):
    name, _, flags = spec.partition(':')
    src = os.path.join(CSRC, 'conv_mfma.hip')
    if flags.startswith('@'):
        path, _, flags = flags[1:].partition(':')
        src = os.path.abspath(path)
    obj, api = f'/tmp/_conv_{name}.o', f'/tmp/_api_{name}.o'
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '-fPIC', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-function', '-Wno-pass-failed',
                           f'-I{CSRC}', *flags.split(), '-c', src, '-o', obj], stderr=subprocess.DEVNULL)
    tag = (os.path.basename(src) + ' ' + flags).strip().replace('"', '')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O2', '-fPIC', '-std=c++17', f'-DMUD_BUILD_FLAGS="{name}: {tag}"', '-c', os.path.join(CSRC, 'api.cpp'), '-o', api])
    lib = os.path.join(OUT, f'lib_{name}.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', *objs, api, obj, '-o', lib])
    return lib


with ThreadPoolExecutor(4) as ex:
    for lib in ex.map(build, sys.argv[1:]):
        print('built', lib)
