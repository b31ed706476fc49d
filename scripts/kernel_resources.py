"""Register / spill / LDS / occupancy table of every k_conv_mfma instantiation, from hipcc's kernel-resource-usage remarks
(no GPU needed):  python scripts/kernel_resources.py [file.hip] [extra hipcc flags]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'mu-diff_amd', 'csrc', 'conv_mfma.hip')
p = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-fPIC', '-std=c++17', '--offload-arch=gfx950', '-Wno-unused-function', '-Wno-pass-failed',
                    '-Rpass-analysis=kernel-resource-usage', *sys.argv[2:], '-c', src, '-o', '/tmp/_kres.o'], stderr=subprocess.PIPE, text=True)
t = p.stderr
if p.returncode:
    sys.exit(t[-3000:])
for b in re.split(r'remark: Function Name: ', t)[1:]:
    name = b.split()[0].strip()
    g = lambda k: re.search(k + r': (\d+)', b).group(1)       # noqa: E731
    m = re.search(r'k_conv_mfmaILi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d)E', name)
    tag = ('KS%s MT%s WM%s WN%s PRO%s DUAL%s PREC%s' % m.groups()) if m else name[:48]
    print('%-48s VGPR %4s AGPR %4s spill %4s scratch %5s occ %s LDS %s' % (tag, g(' VGPRs'), g('AGPRs'), g('VGPRs Spill'), g(r'ScratchSize \[bytes/lane\]'),
                                                                         g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
