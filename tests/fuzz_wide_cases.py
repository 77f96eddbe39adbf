"""The wide random parameter sets shared by the golden generator (tests/golden/make_golden_fuzz.py, oracle, build
container) and the GPU test (tests/test_gpu_fuzz_wide.py, HIP path): same seed -> same sets -> same synthetic inputs."""
import random
from common import random_run_config_wide, warp_trace


def wide_cases(seed, nsets):
    rng = random.Random(seed)
    out = []
    for i in range(nsets):
        kw, data, maxin, npl = random_run_config_wide(rng)
        if i % 4 == 3 and not data.get("warp"):      # every fourth set with badly aligned trace blocks
            data["warp"] = ((rng.choice([3, 5]), rng.choice([300, 580, 900]), 2000) if data["tspace"] > 125
                            else (rng.choice([2, 3, 5]), rng.choice([60, 115, 150])))
        out.append((kw, data, maxin, npl))
    return out


def make_wide_case(data, maxin, npl, pile_select):
    from daccord_amd.synth import SynthData
    d = SynthData(data["genome_len"], data["nreads"], data["read_len"],
                  **{k: v for k, v in data.items() if k not in ("genome_len", "nreads", "read_len", "profile", "warp")})
    prof = tuple(data["profile"]) if data.get("profile") else d.error_profile()
    ovl, piles = pile_select(d.ovl, d.piles, trace_bytes=d.trace_bytes, maxinput=maxin)
    n = min(len(piles), npl)
    trace = d.trace
    if data.get("warp"):
        trace = warp_trace(ovl, piles, d.trace, range(n), *data["warp"])
    return d, prof, ovl, piles, piles[:n], trace
