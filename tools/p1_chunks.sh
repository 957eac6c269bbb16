# per-chunk kernel times of a traced one-column pass for each tools/bin/libkgwas_<name>.so named
for n in "$@"; do echo "== $n"; KGWAS_LIB=$PWD/tools/bin/libkgwas_$n.so timeout 200 python tools/p1_trace.py 2>&1 | awk '/==== traced pass/{p=1} p' | grep "chunk rows=\|==== feed" | cut -c1-90; done
