# dev: one GPU pass of the persistent-generator form (COGAPS_PERSIST): parity on the hardware, bench A/B; PHASE=1 adds the profile build's phase log
mkdir -p gpurun_out/${TAG:-r5_persist}
O=gpurun_out/${TAG:-r5_persist}
export TMPDIR=/tmp
( COGAPS_PERSIST=on timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "headline_shape_stepwise or chained_equals_two" 2>&1 | tail -n 5 ) > $O/on_tests.log
if [ -n "$PHASE" ]; then ( COGAPS_PERSIST=on timeout 300 python tools/dev_chain_log.py 120 > $O/phase_log_persist.txt 2> $O/phase_log_persist.err ); fi
for i in 1 2; do
( timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>$O/base.err | tail -n 1 ) > $O/base_$i.json
( COGAPS_PERSIST=on timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>$O/on.err | tail -n 1 ) > $O/on_$i.json
done
for f in base_1 on_1 base_2 on_2; do python tools/bench_brief.py $O/$f.json; done > $O/ab.txt 2>&1
cat $O/on_tests.log $O/ab.txt
if [ -n "$PHASE" ]; then head -8 $O/phase_log_persist.txt; tail -n 9 $O/phase_log_persist.txt; tail -n 3 $O/phase_log_persist.err; fi
