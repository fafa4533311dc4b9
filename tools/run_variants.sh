# run tools/quick_bench.py with each prebuilt library variant in variants/ (experiments; the default build is restored at the end)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cp pangea-plus_amd/lib/libpangea_hip.so /tmp/lib_default.so
for f in variants/lib_*.so; do
  cp $f pangea-plus_amd/lib/libpangea_hip.so
  echo "== $f"
  timeout 300 python tools/quick_bench.py ${1:-10000000} 3 2>&1 | tail -1
done
cp /tmp/lib_default.so pangea-plus_amd/lib/libpangea_hip.so
