"""Readers for the committed golden vectors (tests/golden/*.npz)."""
from __future__ import annotations

import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def load_network(name: str):
    """The pack that holds golden network `name`: network.npz or network_large.npz."""
    for pack in ("network", "network_large"):
        z = load(pack)
        if f"{name}/logits" in z.files:
            return z
    raise KeyError(name)


def games(pack: str) -> dict[str, dict[str, np.ndarray]]:
    z = load(pack)
    out: dict[str, dict[str, np.ndarray]] = {}
    for key in z.files:
        g, field = key.split("/")
        out.setdefault(g, {})[field] = z[key]
    return out


RULE_PACKS = ("rules_scripted", "rules_5x5", "rules_9x9", "rules_19x19", "rules_edge")


def all_rule_games():
    for pack in RULE_PACKS:
        for name, rec in games(pack).items():
            yield f"{pack}:{name}", rec


def search_cases():
    z = load("search")
    cases = json.loads(bytes(z["cases_json"]).decode())
    for c in cases:
        k = c["name"]
        c["root_actions"] = z[f"{k}/root_actions"]
        c["visits"] = z[f"{k}/visits"]
        c["root_value"] = float(z[f"{k}/root_value"][0])
        c["pi"] = z[f"{k}/pi"]
        c["action"] = int(z[f"{k}/action"][0])
        c["noise_vec"] = z[f"{k}/noise"]
        c["nn_rows"] = int(z[f"{k}/nn_rows"][0])
    return cases


def unpack_planes(packed: np.ndarray, hist: int, n: int) -> np.ndarray:
    nb = (2 * hist + 3) * n * n
    return np.unpackbits(packed)[:nb].reshape(2 * hist + 3, n, n)


def unpack_legal(packed: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(packed)[: n * n + 1]
