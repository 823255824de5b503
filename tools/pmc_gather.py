"""Reduce the counter passes of tools/pmc_gather.sh: per access shape of tools/ubench/gather.hip, what each counter reports against the bytes the
kernel asked for (segments x segment bytes) and against the 64- / 128-byte lines those segments touch.  usage: pmc_gather.py <pass dir> ..."""
import csv, glob, sys, collections
SEGS = 1 << 22
# shape -> (algorithmic bytes, distinct 64-byte lines touched per segment, distinct 128-byte lines per segment); packed 208-byte rows start at any multiple of 16
def lines(seg, gran, stride=None):
    if stride is None:
        return (seg + gran - 1) // gran            # aligned to a 256-byte granule
    import math
    # mean over the row's start offsets modulo the line size
    offs = range(0, gran, math.gcd(stride, gran))
    return sum(((o + seg - 1) // gran) - (o // gran) + 1 for o in offs) / len(offs)
SHAPES = collections.OrderedDict([
    ("gather_seg<4, false>", (4, 1, 1)), ("gather_seg<8, false>", (8, 1, 1)), ("gather_seg<16, false>", (16, 1, 1)),
    ("gather_seg<208, false>", (208, lines(208, 64), lines(208, 128))), ("gather_seg<208, true>", (208, lines(208, 64, 208), lines(208, 128, 208))),
    ("stream16", (None, None, None))])
vals = collections.defaultdict(dict)      # kernel -> counter -> value of its LAST dispatch
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        last = {}
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            last[(name, r["Counter_Name"])] = (int(r["Dispatch_Id"]), float(r["Counter_Value"]))
        for (name, c), (_, v) in last.items():
            vals[name][c] = v
counters = sorted({c for v in vals.values() for c in v})
print("tools/ubench/gather.hip under rocprofv3 --pmc (separate passes), MI355X; %d segments per gather kernel at random addresses of a 4 GiB buffer (each once);" % SEGS)
print("stream16 = 1 GiB read at 16 B per lane, consecutive.  FETCH_SIZE is in KiB as the tool prints it (no correction applied here).")
print()
print("%-26s %14s %14s %14s | %s" % ("kernel", "asked bytes", "64-B lines", "128-B lines", "  ".join("%22s" % c for c in counters)))
for k, (seg, l64, l128) in SHAPES.items():
    name = next((n for n in vals if n.startswith(k)), None)
    if name is None:
        print("%-26s (no record)" % k); continue
    asked = (1 << 30) if seg is None else seg * SEGS
    n64 = asked / 64 if seg is None else l64 * SEGS
    n128 = asked / 128 if seg is None else l128 * SEGS
    print("%-26s %14.0f %14.0f %14.0f | %s" % (k, asked, n64, n128, "  ".join("%22.0f" % vals[name].get(c, float("nan")) for c in counters)))
print()
print("ratios: FETCH_SIZE x 1024 / asked bytes;  FETCH_SIZE x 1024 / (64 B x lines touched);  RDREQ / 64-B lines;  RDREQ_32B / RDREQ")
for k, (seg, l64, l128) in SHAPES.items():
    name = next((n for n in vals if n.startswith(k)), None)
    if name is None: continue
    asked = (1 << 30) if seg is None else seg * SEGS
    n64 = asked / 64 if seg is None else l64 * SEGS
    v = vals[name]
    fs = v.get("FETCH_SIZE", float("nan")) * 1024.0
    rd = v.get("TCC_EA0_RDREQ_sum", float("nan")); rd32 = v.get("TCC_EA0_RDREQ_32B_sum", float("nan"))
    print("%-26s fetch/asked %6.3f   fetch/(64 B x lines) %6.3f   rdreq/lines %6.3f   rdreq_32B/rdreq %6.3f" % (k, fs / asked, fs / (64.0 * n64), rd / n64, rd32 / rd if rd else float("nan")))
