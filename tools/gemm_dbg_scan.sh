for cfg in 8 4; do echo "cfg=$cfg"; ASR_GEMM_CFG=$cfg python tools/gemm_bench.py nt 2>&1 | grep -v amdgpu | cut -c1-72; done
