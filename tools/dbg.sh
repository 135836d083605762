for d in 0 1 2 4 8 3 6 7 15; do echo "== DEBUG $d"; LWDETR_GEMM_DEBUG=$d python tools/gemm_bench.py 2>&1 | grep -E "^(qkv|proj|fc1-noact|fc2) "; done
