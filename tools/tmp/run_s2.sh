set -x
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "dwconv" 2>&1 | tail -5
for sh in 2,513,1025,128 2,257,513,256 2,129,257,728; do timeout 120 python tools/lab/op_time.py dw_fwd_s2 --shape $sh; done
timeout 300 python bench.py --steps 30 --warmup 5 2>&1 | tail -1
