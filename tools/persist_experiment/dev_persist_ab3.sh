# dev: A/B of the persistent-generator forms -- MODES="off seq seq:noloop each each:noloop on"
mkdir -p gpurun_out/${TAG:-r5_persist_ab}; O=gpurun_out/${TAG:-r5_persist_ab}
for m in ${MODES:-off seq seq:noloop each each:noloop on}; do
  pm=${m%%:*}; nl=""; [ "$m" != "$pm" ] && nl=1
  if [ -n "$nl" ]; then export COGAPS_PERSIST_NOLOOP=1; else unset COGAPS_PERSIST_NOLOOP; fi
  ( COGAPS_PERSIST=$pm timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 2>$O/$m.err | tail -n 1 ) > $O/$m.json; echo $m; python tools/bench_brief.py $O/$m.json
done > $O/ab.txt 2>&1
cat $O/ab.txt
