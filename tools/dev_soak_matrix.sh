cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r6_soak2
cat > /tmp/one.py <<'PY'
import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import dev_soak, bench
data = bench.synthetic_dense(20000, 2000)
r = dev_soak.run(data, 100, int(sys.argv[1]), dict(nPatterns=50, seed=42, outputFrequency=10), foreign_seconds=float(sys.argv[2]) if float(sys.argv[2]) > 0 else None)
print(json.dumps({k: r[k] for k in ("seconds", "proposals", "state_digest", "foreign") if k in r}), {w: (r[w]["chained"], r[w]["recoveries"]) for w in "AP"})
PY
for cfg in "quiet:0:" "chain:${SOAK_SECS:-40}:" "nochain:${SOAK_SECS:-40}:COGAPS_NO_CHAIN=1" "nograph:${SOAK_SECS:-40}:COGAPS_NO_GRAPH=1" "nochain_nograph:${SOAK_SECS:-40}:COGAPS_NO_CHAIN=1 COGAPS_NO_GRAPH=1"; do
  name=${cfg%%:*}; rest=${cfg#*:}; secs=${rest%%:*}; envs=${rest#*:}
  echo "== $name"; env $envs timeout 300 python /tmp/one.py ${SOAK_ITERS:-30} $secs 2>&1 | grep -v "amdgpu.ids" | tail -3
done
