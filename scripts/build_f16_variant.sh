# experiment build: the 16-bit pieces of the 16x3 plan as fp16 (11-bit) instead of bf16 (8-bit) - conv kernels + weight packing only
set -e
cd /root/repo/mu-diff_amd/csrc
make -j8 > /dev/null
mkdir -p ../mudiff_hip/variants
sed 's/^typedef __bf16 h16;/typedef _Float16 h16;/' conv_mfma.hip > _conv_f16_tmp.hip
grep -q "typedef _Float16 h16;" _conv_f16_tmp.hip
/opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -c _conv_f16_tmp.hip -o /tmp/_conv_f16.o
/opt/rocm/bin/hipcc -O2 -fPIC -std=c++17 -DMUD_BUILD_FLAGS='"h16=fp16"' -c api.cpp -o /tmp/_api_f16.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC elementwise.o groupnorm.o dense.o conv_direct.o fir.o attention.o /tmp/_api_f16.o /tmp/_conv_f16.o -o ../mudiff_hip/variants/lib_f16.so
rm -f _conv_f16_tmp.hip
echo built variants/lib_f16.so
