cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/dev_ab.sh "ab_libs/libE_erasebins.so ab_libs/libG_statictree.so" --steps 20 --warmup 5
for g in 256 320 384 512; do COGAPS_FUSED_GRID=$g python bench.py --no-cpu --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']
print('grid $g %8d  evalA %.2f evalP %.2f genA %.2f' % (round(d['value']), k[0]['avg_launch_us'], k[1]['avg_launch_us'], k[2]['avg_launch_us']))"; done
