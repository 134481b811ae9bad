"""Error budget of bf16 inference on the golden (seed-generated) networks, on the CPU: distance from the float32
module of the emulation with (a) bf16 weights only, (b) + bf16 activations with a float32 skip connection,
(c) + bf16 skip connection. Writes profiles/r02_bf16_error_budget.json when given --out."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import bf16_emul  # noqa: E402
import stubnets  # noqa: E402
from tests_common import GOLDEN_NETS, build_golden_net  # noqa: E402

torch.set_num_threads(8)
out = {}
for name in GOLDEN_NETS:
    net, _, kw = build_golden_net(name)
    c = 2 * kw.get("history_len", 8) + 5
    x = torch.from_numpy(stubnets.synthetic_planes(37, c, kw["board_size"], seed=5))
    with torch.inference_mode():
        lg_t, v_t = net(x)
    row = {"logit_scale": float(lg_t.abs().max())}
    for label, kwargs in [("input_only", dict(weights=False, activations=False)),
                          ("weights", dict(weights=True, activations=False)),
                          ("weights+act(f32 skip, f32 se image)", dict(weights=True, activations=True, residual="f32", se_image=False)),
                          ("weights+act(f32 skip)", dict(weights=True, activations=True, residual="f32")),
                          ("weights+act(bf16 skip)", dict(weights=True, activations=True, residual="bf16")),
                          ("act only (f32 skip)", dict(weights=False, activations=True, residual="f32")),
                          ("act only (bf16 skip)", dict(weights=False, activations=True, residual="bf16"))]:
        lg, v = bf16_emul.emulate(net, x, **kwargs)
        e = bf16_emul.errors(lg, v, lg_t, v_t)
        row[label] = dict(prob=e[0], value=e[1], logit=e[2])
    out[name] = row
    print(name, json.dumps(row, indent=1), flush=True)
if "--out" in sys.argv:
    with open(os.path.join(ROOT, "profiles", "r02_bf16_error_budget.json"), "w") as f:
        json.dump(out, f, indent=1)
