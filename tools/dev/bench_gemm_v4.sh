cd $GRAFT_REPO_ROOT
echo "pp64:"; CROPS=337 python tools/bench_gemm_epi.py 2>&1 | grep -v amdgpu
echo "k_gemm_f16 (2 WG/CU):"; VILGOD_HIP_LIB=$PWD/vilgod_amd/libvilgod_hip_dev.so VG_GEMM_V4=1 CROPS=337 python tools/bench_gemm_epi.py 2>&1 | grep -v amdgpu
