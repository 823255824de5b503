"""dev: distribution of the phases of the chained launch's generator workgroup over thousands of launches (profile build, chip-wide 100 MHz clock):
entry -> first barrier -> ahead-of-the-decisions work done -> granules in -> decisions applied + barrier -> attempt wave 0 ends; by number of rounds."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
PL = _capi.bind(ctypes.CDLL(os.environ.get('COGAPS_PROFILE_LIB', os.path.join(os.path.dirname(_capi.LIB_PATH), 'libcogaps_hip_PROFILE_DEV.so'))))      # (COGAPS_PROFILE_LIB: another profile build, for A/B phase logs)
SPARSE = '--sparse' in sys.argv
if SPARSE:      # BASELINE configs[4]'s shard shape, as bench.py --sparse --genes 50000 --samples 12500 makes it
    sys.argv.remove('--sparse')
    data = synthetic_dense(50000, 12500)
    data = (data * (np.random.Generator(np.random.MT19937(777)).random(data.shape) >= 0.95)).astype(np.float32)
    S = _capi.Session(data, lib=PL, nPatterns=50, nIterations=100, seed=42, sparseOptimization=True)
else:
    data = synthetic_dense(20000, 2000)
    S = _capi.Session(data, lib=PL, nPatterns=50, nIterations=100, seed=42)
S.run_iterations(1, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 60)
N = 65536
buf = (ctypes.c_uint64 * (N * 8))(); n = ctypes.c_uint32()
PL.cogaps_debug_chain_log.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert PL.cogaps_debug_chain_log(buf, ctypes.byref(n)) == 0
a = np.array(buf, dtype=np.int64).reshape(N, 8)[:min(n.value, N)]
a = a[a[:, 0] > 0]
t = (a[:, 1:6] - a[:, 0:5]) / 100.0            # phase lengths, us
tot = (a[:, 5] - a[:, 0]) / 100.0
prevq = a[:, 6] & 0xFFFF; em = (a[:, 6] >> 16) & 0xFFFF; spec = (a[:, 6] >> 32) & 1; rounds = a[:, 7]
level = (a[:, 6] >> 36) & 3; bad = (a[:, 6] >> 40) & 0xFF; vj = ((a[:, 6] >> 48) & 0xFFFF) / 100.0      # lanes that drew again; applied -> validated + joined with the flush, us
names = ['entry -> first barrier', 'first barrier -> window drawn ahead (attempt lane 0)', 'drawn ahead -> granules in (helper lane 0; may be negative)', 'granules in -> applied + barrier', 'round(s) of the next batch']
def line(name, v): print('  %-32s mean %6.2f  p10 %6.2f  p50 %6.2f  p75 %6.2f  p90 %6.2f  p99 %6.2f' % (name, v.mean(), *np.percentile(v, [10, 50, 75, 90, 99])))
print('%d launches logged (queue >= 100); rounds: %s; classified ahead: %.3f' % (len(a), dict(zip(*np.unique(rounds, return_counts=True))), spec.mean()))
for sel, nm in ((rounds >= 1, 'all'), (rounds == 1, 'one round'), (rounds == 2, 'two rounds')):
    if sel.sum() < 10: continue
    print(nm, '(%d)' % sel.sum())
    line('whole workgroup', tot[sel])
    for i, n_ in enumerate(names): line(n_, t[sel, i])
one = rounds == 1
r = t[one, 4]
print('one-round launches, the round by quartile of its length: queue %s erased %s' % (
    [round(float(prevq[one][(r >= lo) & (r <= hi)].mean()), 1) for lo, hi in zip(np.percentile(r, [0, 25, 50, 75]), np.percentile(r, [25, 50, 75, 100]))],
    [round(float(em[one][(r >= lo) & (r <= hi)].mean()), 1) for lo, hi in zip(np.percentile(r, [0, 25, 50, 75]), np.percentile(r, [25, 50, 75, 100]))]))
w = t[:, 2]
for lo, hi in ((100, 200), (200, 240), (240, 256), (256, 2000)):
    sel = (prevq > lo) & (prevq <= hi)
    if sel.sum(): print('  queue in (%d, %d]: %5d launches, wait for the granules mean %.2f p50 %.2f p90 %.2f; apply mean %.2f' % (lo, hi, sel.sum(), w[sel].mean(), np.median(w[sel]), np.percentile(w[sel], 90), t[sel, 3].mean()))

print('lanes that drew again per launch: mean %.2f; launches with none %.3f; with erased atoms %.3f' % (bad.mean(), (bad == 0).mean(), (em > 0).mean()))
for nm, sel in (('no lane drew again, nothing erased', (bad == 0) & (em == 0) & one), ('no lane drew again, atoms erased', (bad == 0) & (em > 0) & one), ('some lane drew again', (bad > 0) & one),
                ('... keeping their picks (level 1)', (level == 1) & one), ('... behind the flush (level 2)', (level == 2) & one)):
    if sel.sum() > 5: print('  one round, %-36s %5d launches: whole %.2f  round %.2f  validate + join %.2f us' % (nm, sel.sum(), tot[sel].mean(), t[sel, 4].mean(), vj[sel].mean()))

why = (ctypes.c_uint64 * 8)()
PL.cogaps_debug_ahead_why.argtypes = [ctypes.c_void_p]
if PL.cogaps_debug_ahead_why(why) == 0:
    w = [int(x) for x in why]
    if w[0]:
        print('lanes drawn ahead (whole run): %d; drew again %.4f; by cause (a lane may have several): pick moved with the domain\'s size %.4f, slot refilled by the flush %.5f, '
              'noted atom record %.4f, noted matrix cell %.4f, a birth\'s bitmap words %.4f, a long way (walk / full search / front()) %.4f'
              % (w[0], w[1] / w[0], w[2] / w[0], w[3] / w[0], w[4] / w[0], w[5] / w[0], w[6] / w[0], w[7] / w[0]))
