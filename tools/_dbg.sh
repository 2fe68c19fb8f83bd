for i in 1 2; do
echo "== new"; timeout 200 python tools/kbench.py --only gemm 2>&1 | grep '"nt' | grep -v hybrid
echo "== old"; timeout 200 python _ab_old/tools/kbench.py --only gemm 2>&1 | grep '"nt'
done
