#!/usr/bin/env python3
"""Per-block max relative error of the HIP path against the golden fixtures (both state layouts) -- the numbers
quoted in DESIGN.md section 2.  Needs a GPU; reads only tests/golden/*.npz."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import golden_util as gu  # noqa: E402
from tests import gpu_util as G  # noqa: E402

out = {}
for name, per_env in (("g2_hummingbird_raw", False), ("g3_crazyflie", False), ("g4_randomized", True)):
    d = gu.load(name)
    blocks = gu.env_blocks(d)
    for alias in (0, 1):
        n = max(64, len(blocks))
        b0 = blocks[0]
        if per_env:
            rows = np.stack([G.model_row(gu.sub(blocks[i % len(blocks)], "const_")) for i in range(n)])
            h = G.Handle(n, float(b0["dt"]), int(b0["sim_steps"]), int(b0["ep_len"]), rows=rows, alias=alias)
        else:
            const = gu.sub(d, "const_") if any(k.startswith("const_") for k in d) else gu.sub(b0, "const_")
            h = G.Handle(n, float(b0["dt"]), int(b0["sim_steps"]), int(b0["ep_len"]), const=const, alias=alias)
        outs, _ = G.run_blocks(h, blocks, n)
        errs = [gu.rel_err(o["obs"], b["obs"]) for o, b in zip(outs, blocks)]
        out["%s/%s" % (name, "alias" if alias else "plain")] = {"max": max(errs), "median": float(np.median(errs)),
                                                               "blocks": len(errs), "steps": int(blocks[0]["obs"].shape[0]),
                                                               "worst_block": int(np.argmax(errs)),
                                                               "all": [float("%.3g" % e) for e in errs]}
        h.close()
print(json.dumps(out, indent=1))
