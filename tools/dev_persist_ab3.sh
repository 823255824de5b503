mkdir -p gpurun_out/r5_persist7; O=gpurun_out/r5_persist7
( COGAPS_PERSIST=each timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline_shape_stepwise" 2>&1 | tail -n 3 ) > $O/each_tests.log
for m in off each on each; do
( COGAPS_PERSIST=$m timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>$O/$m.err | tail -n 1 ) > $O/$m.json; echo $m; python tools/bench_brief.py $O/$m.json
done > $O/ab.txt 2>&1
cat $O/each_tests.log $O/ab.txt
