#!/usr/bin/env python3
"""Rules edge cases for tests/golden/rules_edge.npz, produced by running the upstream Python implementation
(/root/reference, read-only) -- build container only:  python tests/golden/make_golden_edge.py

What the pack adds to rules_*.npz (same record layout, see make_golden.record_game):
  * snapback: a move captures ONE stone and the opponent's reply on the captured point takes the capturing group
    (>= 2 stones) -- legal, because the position is new (go.py:122-143);
  * superko cycles longer than simple ko: points that are illegal because the resulting situation was seen MORE than
    two plies ago (go.py:138-141), and board repetitions with the other side to move, which situational superko allows;
  * the move cap: games that end at move_count == 2 n^2 without two passes (go.py:205-206), 5x5 and 9x9;
  * passes that repeat a situation (a pass performs no superko check, go.py:162-176) followed by stone moves that the
    enlarged seen set forbids;
  * scoring: an empty region bordering both colours counts for nobody, a region bordering nobody (empty board) too
    (go.py:208-240).
Games are found by seeded random search under a policy that steers towards these events; only the action lists and
the upstream outputs are stored. MANIFEST_edge.json records the event counts; the generator fails if they fall
short of the floor asserted at the end.
"""
from __future__ import annotations

import json
import os
import random
import sys
from concurrent.futures import ProcessPoolExecutor

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np

import make_golden as mg          # imports the upstream package from /root/reference
from make_golden import GoState, decode_action


def _hash(board, to_play):
    return GoState._state_hash(board, to_play)


def superko_census(st, first_seen, ply):
    """(points illegal by superko, those whose repeated situation lies more than two plies back) of position `st`,
    which is `ply` plies into a game whose situations were first seen at the plies in `first_seen`."""
    n = st.board_size
    pts = far = 0
    for r in range(n):
        for c in range(n):
            if st.board[r, c] != 0 or st.is_legal((r, c)):
                continue
            b = st.board.copy()
            b[r, c] = st.to_play
            st._remove_dead_groups_around(b, r, c, -st.to_play)
            _, libs = st._group_and_liberties(b, r, c)
            if libs:
                pts += 1
                if ply + 1 - first_seen[_hash(b, -st.to_play)] > 2:
                    far += 1
    return pts, far


def analyse(n, hist, komi, actions):
    """Replay `actions` upstream and count the events of interest."""
    st = GoState.new_game(board_size=n, komi=komi, history_len=hist)
    first_seen = {_hash(st.board, st.to_play): 0}
    boards_seen = {st.board.tobytes(): {st.to_play}}
    ev = dict(superko_points=0, long_cycle_points=0, positional_repeats=0, snapbacks=0, pass_repeats=0,
              moves=0, cap_end=0, pass_end=0)
    prev_cap_point, prev_cap_count = None, 0
    for ply, a in enumerate(actions):
        pts, far = superko_census(st, first_seen, ply)
        ev["superko_points"] += pts
        ev["long_cycle_points"] += far
        move = decode_action(n, a)
        stones_before = int((st.board != 0).sum())
        nxt = st.play(move)
        captured = stones_before + (0 if move is None else 1) - int((nxt.board != 0).sum())
        if move is not None and prev_cap_point == move and prev_cap_count == 1 and captured >= 2:
            ev["snapbacks"] += 1
        if move is not None and captured == 1:
            diff = np.argwhere((st.board != 0) & (nxt.board == 0))
            prev_cap_point, prev_cap_count = (int(diff[0][0]), int(diff[0][1])), 1
        else:
            prev_cap_point, prev_cap_count = None, 0
        st = nxt
        h = _hash(st.board, st.to_play)
        if h in first_seen:
            if move is None:
                ev["pass_repeats"] += 1
        else:
            first_seen[h] = ply + 1
        sides = boards_seen.setdefault(st.board.tobytes(), set())
        if move is not None and sides and st.to_play not in sides:
            ev["positional_repeats"] += 1
        sides.add(st.to_play)
    ev["moves"] = len(actions)
    if st.is_terminal():
        if st.consecutive_passes >= 2:
            ev["pass_end"] = 1
        else:
            ev["cap_end"] = 1
    return ev


def steered_game(args):
    """One seeded game under a policy that prefers (in this order) a snapback recapture, a move that repeats a board
    with the other side to move, a move that leaves the opponent superko-forbidden points (old repetitions count
    double), a capture; otherwise uniform over the legal stone moves; passes with `pass_prob`."""
    n, hist, komi, seed, pass_prob = args
    rng = random.Random(seed)
    st = GoState.new_game(board_size=n, komi=komi, history_len=hist)
    boards_seen = {st.board.tobytes(): {st.to_play}}
    first_seen = {_hash(st.board, st.to_play): 0}
    greedy = n <= 5 and seed % 3 != 0
    acts = []
    last_single_capture = None
    while not st.is_terminal():
        legal = st.legal_actions()
        stones = legal[:-1]
        a = None
        if not stones or rng.random() < pass_prob:
            a = legal[-1]
        else:
            options = []
            for s in stones:
                mv = decode_action(n, s)
                nb = st.play(mv)
                cap = int((st.board != 0).sum()) + 1 - int((nb.board != 0).sum())
                sides = boards_seen.get(nb.board.tobytes(), set())
                score = 0
                if last_single_capture == mv and cap >= 2:
                    score = 100
                elif sides and nb.to_play not in sides:
                    score = 50
                elif cap > 0:
                    score = 1
                if greedy:
                    fs = first_seen if _hash(nb.board, nb.to_play) in first_seen else {**first_seen, _hash(nb.board, nb.to_play): len(acts) + 1}
                    pts, far = superko_census(nb, fs, len(acts) + 1)
                    score += 4 * pts + 8 * far
                options.append((score, s, cap))
            best = max(o[0] for o in options)
            pool = [o for o in options if o[0] == best] if (best >= 2 and rng.random() < 0.85) or rng.random() < 0.5 else options
            a = rng.choice(pool)[1]
        mv = decode_action(n, a)
        nb = st.play(mv)
        cap = int((st.board != 0).sum()) + (0 if mv is None else 1) - int((nb.board != 0).sum())
        if mv is not None and cap == 1:
            d = np.argwhere((st.board != 0) & (nb.board == 0))
            last_single_capture = (int(d[0][0]), int(d[0][1]))
        else:
            last_single_capture = None
        st = nb
        boards_seen.setdefault(st.board.tobytes(), set()).add(st.to_play)
        acts.append(a)
        first_seen.setdefault(_hash(st.board, st.to_play), len(acts))
    return (n, hist, komi, seed, pass_prob), acts, analyse(n, hist, komi, acts)


def beam_game(args):
    """Beam search over whole games for lines with many superko-forbidden points (old repetitions weighted x5):
    every ply each kept line is extended by every legal stone move (and the pass), the `width` best by cumulative
    census are kept, ties broken by the seeded generator."""
    n, hist, komi, seed, width = args
    rng = random.Random(seed)
    st0 = GoState.new_game(board_size=n, komi=komi, history_len=hist)
    beam = [(0, st0, {_hash(st0.board, st0.to_play): 0}, [])]
    done = []
    while beam:
        nxt = []
        for score, st, fs, acts in beam:
            if st.is_terminal():
                done.append((score, acts))
                continue
            legal = st.legal_actions()
            for a in legal:
                if a == n * n and len(legal) > 1 and rng.random() < 0.7:
                    continue
                nb = st.play(decode_action(n, a))
                h = _hash(nb.board, nb.to_play)
                fs2 = fs if h in fs else {**fs, h: len(acts) + 1}
                pts, far = superko_census(nb, fs2, len(acts) + 1) if not nb.is_terminal() else (0, 0)
                nxt.append((score + pts + 5 * far + rng.random() * 0.5, nb, fs2, acts + [a]))
        nxt.sort(key=lambda t: -t[0])
        beam = nxt[:width]
    done.sort(key=lambda t: -t[0])
    acts = done[0][1]
    return (n, hist, komi, seed, -1.0), acts, analyse(n, hist, komi, acts)


def rc(n, *pairs):
    return [n * n if p is None else p[0] * n + p[1] for p in pairs]


def scripted():
    g = {}
    # an empty region that borders both colours counts for nobody; komi decides (go.py:208-240)
    g["neutral_region"] = (5, 0.5, 8, rc(5, (2, 1), (2, 3), None, None))
    # walls: left two columns border Black only, right two White only
    g["two_territories"] = (5, 5.5, 8, rc(5, (0, 2), (0, 3), (1, 2), (1, 3), (2, 2), (2, 3), (3, 2), (3, 3), (4, 2), (4, 3), None, None))
    # the empty board borders nobody: 0 : komi
    g["empty_board_end"] = (5, 5.5, 8, rc(5, None, None))
    # a stone move after two passes is still accepted by play() (go.py:159-203 does not look at is_terminal)
    g["play_after_terminal"] = (5, 5.5, 4, rc(5, (2, 2), None, None, (1, 1), None))
    return g


def main():
    out = {}
    events = {}
    for name, (n, komi, hist, acts) in scripted().items():
        out[name] = mg.record_game(n, komi, hist, actions=acts)
        events[name] = analyse(n, hist, komi, acts)
    jobs = []
    for n, count, pp in ((3, 300, 0.0), (4, 400, 0.0), (4, 150, 0.08), (5, 300, 0.0), (5, 120, 0.05), (6, 120, 0.0)):
        jobs += [(n, 8 if n != 6 else 3, 5.5 if n != 3 else 0.5, 7000 + 1000 * n + s + int(pp * 1000), pp) for s in range(count)]
    jobs += [(9, 8, 5.5, 9100 + s, 0.0) for s in range(6)]
    beams = [(3, 8, 0.5, 500 + s, 40) for s in range(8)] + [(4, 8, 5.5, 600 + s, 40) for s in range(14)] + \
            [(5, 8, 5.5, 700 + s, 24) for s in range(8)]
    with ProcessPoolExecutor(8) as ex:
        fut = [ex.submit(beam_game, b) for b in beams]
        res = list(ex.map(steered_game, jobs, chunksize=8))
        res += [f.result() for f in fut]

    def pick(key, n, k, need=1, extra=None):
        cand = [r for r in res if r[0][0] == n and r[2][key] >= need and (extra is None or extra(r))]
        cand.sort(key=lambda r: (-r[2][key], len(r[1]), r[0][3]))
        return cand[:k]

    chosen = {}

    def take(tag, rows):
        for r in rows:
            chosen.setdefault((r[0][0], r[0][3], r[0][4]), (tag, r))

    for n in (3, 4, 5, 6):
        take("snapback", pick("snapbacks", n, 2))
        take("longcycle", pick("long_cycle_points", n, 4 if n <= 5 else 1))
        take("posrepeat", pick("positional_repeats", n, 2 if n <= 5 else 1))
        take("superko", pick("superko_points", n, 5 if n <= 5 else 1))
    for r in res:
        if r[0][4] < 0:
            take("beam", [r])
    take("passrepeat", pick("pass_repeats", 4, 2))
    take("passrepeat", pick("pass_repeats", 5, 2))
    take("movecap", pick("cap_end", 5, 2))
    take("movecap", pick("cap_end", 9, 1))
    take("movecap", pick("cap_end", 4, 1))
    for (n, seed, pp), (tag, r) in sorted(chosen.items()):
        name = f"{tag}_n{n}_s{seed}"
        cfg, acts, ev = r
        out[name] = mg.record_game(cfg[0], cfg[2], cfg[1], actions=acts)
        events[name] = ev
    flat = {}
    for gname, rec in out.items():
        for k, v in rec.items():
            flat[f"{gname}/{k}"] = v
    np.savez_compressed(os.path.join(HERE, "rules_edge.npz"), **flat)
    total = {k: int(sum(e[k] for e in events.values())) for k in next(iter(events.values()))}
    info = dict(games=len(out), totals=total, per_game=events, searched_games=len(jobs))
    with open(os.path.join(HERE, "MANIFEST_edge.json"), "w") as f:
        json.dump(info, f, indent=1)
    print(json.dumps(dict(games=len(out), totals=total), indent=1))
    assert total["superko_points"] >= 300 and total["long_cycle_points"] >= 40 and total["snapbacks"] >= 3, total
    assert total["positional_repeats"] >= 5 and total["cap_end"] >= 3 and total["pass_repeats"] >= 2, total
    for n in (5, 9):
        assert any(e["cap_end"] for g, e in events.items() if f"_n{n}_" in g), f"no move-cap ending on {n}x{n}"


if __name__ == "__main__":
    main()
