"""The emit kernel over image counts around the Infinity-Cache knee, in ONE process on one box, the store
policy and the chain route switched through the debug hooks between sweeps (alternating, twice):
    emit_nt_min_bytes       1 = non-temporal stores always, 10^12 = never, 0 = the library's default
    inline_chain_max_bytes  1 = chain prep + emit always, 10^12 = the emit kernel walks the chain always, 0 = default
python tools/exp/emit_sweep_probe.py [model] [sizes,comma,separated]"""
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
from visgeom_amd import benchlib, capi, synthetic  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "eucm"
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [5000, 8000, 9000, 10000, 11000, 12000, 13000, 15000, 20000, 30000]
d = synthetic.make_mono(model, max(sizes), 1)
NEVER, ALWAYS = 10 ** 12, 1
settings = [("nt=never  route=default", NEVER, 0), ("nt=always route=default", ALWAYS, 0), ("nt=never  route=inline", NEVER, 10 ** 12),
            ("nt=always route=inline", ALWAYS, 10 ** 12), ("nt=never  route=prep", NEVER, 1), ("nt=always route=prep", ALWAYS, 1),
            ("library defaults", 0, 0)]
for rep in range(2):
    for name, nt, inl in settings:
        capi.debug_set("emit_nt_min_bytes", nt)
        capi.debug_set("inline_chain_max_bytes", inl)
        for r in benchlib.emit_sweep(d, model, sizes, reps=60):
            print("%-24s %6d images %7.1f MB  %-12s kernel %7.2f us (%.3f)  step %7.2f us (%.3f)" % (
                name, r["images"], r["output_MB"], r["route"], r["kernel_us"], r["frac"], r["step_us"], r["frac_whole_step"]), flush=True)
