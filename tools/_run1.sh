timeout 900 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_lat_kernel.py tests/test_gpu_parity.py tests/test_gpu_lean.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
BN_VARIANT=timing timeout 300 python tools/stamps_blog.py 2>&1 | grep "chain at" | cut -c1-300
for i in 1 2 3; do
for v in c2 main; do
echo -n "$v  "; BN_TOOL_LIB=$v timeout 300 python tools/region_overhead.py 2>&1 | grep "overlap=True" | sed 's/.*K=50:/K=50:/'
done
done
