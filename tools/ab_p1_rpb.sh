for r in 0 512 768 1024 1280; do if [ $r = 0 ]; then unset KGWAS_COARSE_RPB; else export KGWAS_COARSE_RPB=$r; fi; python tools/p1_large_once.py 100000000 6 2>&1 | grep -v amdgpu; done
unset KGWAS_COARSE_RPB; python tools/p1_large_once.py 1200000000 4 2>&1 | grep -v amdgpu
