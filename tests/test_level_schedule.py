"""The level pipeline's schedule (u-rnn_amd/level_schedule.py) through a happens-before model: CPU only, no HIP.

RolloutEngine issues exactly the event lists of ``level_schedule.events`` on its streams; here the same lists drive vector clocks (one
component per stream, a launch ticks its stream, ``wait(dst, src)`` merges src's clock into dst's) and every buffer access of every launch
(``level_schedule.ACCESS``) is checked: a read of frame s from a ring slot must be ordered AFTER the write of frame s to that slot and every
other write to the slot must be ordered before that write or after the read; writes to one slot must be totally ordered.  A schedule that
passes cannot race whatever the kernels' timing -- the bit-for-bit GPU tests (test_hip_rollout.py) only ever see one timing."""
import itertools

import pytest

from urnn_amd import level_schedule as ls


class Model:
    def __init__(self, plan):
        self.plan = plan
        self.n = ls.STREAMS + 1                                  # index 0: CUR
        self.vc = [[0] * self.n for _ in range(self.n)]
        self.writes = {}                                         # (buffer, slot) -> [(frame, stream index, clock)]
        self.reads = {}                                          # (buffer, slot) -> [(frame wanted, clock, launch)]
        self.errors = []
        self.first_frame = 0

    def slots(self, buf):
        r = ls.RINGS[buf]
        return self.plan["period"] if r == "period" else r

    @staticmethod
    def hb(w_stream, w_clock, clock):
        """did the launch with clock w_clock on stream w_stream happen before a point with vector clock ``clock``?"""
        return w_clock[w_stream] <= clock[w_stream]

    def run(self, events):
        for ev in events:
            if ev[0] == "wait":
                d, s = ev[1] + 1, ev[2] + 1
                self.vc[d] = [max(a, b) for a, b in zip(self.vc[d], self.vc[s])]
                continue
            _, q, u, name, tau = ev
            s = q + 1
            self.vc[s][s] += 1
            clock = list(self.vc[s])
            reads, writes = ls.ACCESS[name]
            for buf, off in reads:
                want = tau + off
                key = (buf, want % self.slots(buf))
                ws = self.writes.get(key, [])
                src = [w for w in ws if w[0] == want]
                if want < self.first_frame:                      # a state of before the event: the slot reset() zeroed -- nobody may have written it
                    bad = [w for w in ws]
                    if bad:
                        self.errors.append(f"{name}({tau}) wants the initial {buf} but frame {bad[0][0]} was written to its slot")
                elif not src:
                    self.errors.append(f"{name}({tau}) reads {buf}[{want}] which nobody has been asked to write yet")
                else:
                    f, wst, wcl = src[-1]
                    if not self.hb(wst, wcl, clock):
                        self.errors.append(f"{name}({tau}) reads {buf}[{want}] without waiting for its writer")
                    for f2, wst2, wcl2 in ws:
                        if f2 != want and not self.hb(wst2, wcl2, wcl):   # a different frame in the slot, not ordered before the wanted write
                            self.errors.append(f"{name}({tau}) reads {buf}[{want}] but frame {f2} may already sit in the slot")
                self.reads.setdefault(key, []).append((want, clock, s, f"{name}({tau})"))
            for buf, off in writes:
                frame = tau + off
                key = (buf, frame % self.slots(buf))
                for f2, wst2, wcl2 in self.writes.get(key, []):
                    if not self.hb(wst2, wcl2, clock):
                        self.errors.append(f"{name}({tau}) writes {buf}[{frame}] unordered against the write of frame {f2}")
                for want, rcl, rs, who in self.reads.get(key, []):
                    if want != frame and not (rcl[rs] <= clock[rs]):
                        self.errors.append(f"{name}({tau}) overwrites {buf}[{want}] with frame {frame} while {who} may still be reading it")
                self.writes.setdefault(key, []).append((frame, s, clock))


def check_runs(plan, pieces, graphs, whole=False):
    m = Model(plan)
    f = 0
    for n in pieces:
        for key, its in ls.replays(plan, f, n, graphs=graphs, whole=whole):
            m.run(ls.events(plan, its))
        f += n
    return m


PIECES = [(1,), (2,), (3,), (4,), (5,), (7,), (13,), (30,), (36,), (3, 7, 1, 12), (6, 6, 11), (1, 1, 2, 5, 6, 8), (10, 13), (4, 4, 4, 4, 4, 4), (29, 31)]


@pytest.mark.parametrize("name", sorted(ls.PLANS))
@pytest.mark.parametrize("graphs", [True, False, "whole"])
def test_no_launch_can_race_on_a_ring_slot(name, graphs):
    """graphs: the run cut into fill / group / steady / drain replays | one eager pass | one replay per run() call (the whole-run graphs)"""
    plan = ls.PLANS[name]
    for pieces in PIECES:
        m = check_runs(plan, pieces, bool(graphs), whole=graphs == "whole")
        assert not m.errors, f"plan {name}, run() calls of {pieces}: " + "; ".join(m.errors[:4])
        done = sum(pieces)
        # every frame went through every launch exactly once
        assert all(len([w for w in ws if True]) >= 1 for ws in m.writes.values())
        frames = sorted(w[0] for w in itertools.chain.from_iterable(v for k, v in m.writes.items() if k[0] == "feat"))
        assert frames == list(range(done)), f"plan {name} {pieces}: the head's feature map was written for frames {frames}"


@pytest.mark.parametrize("name", sorted(ls.PLANS))
def test_units_cover_the_network_in_order(name):
    plan = ls.PLANS[name]
    seen = {}
    for st, lag, names in plan["units"]:
        assert 0 <= st < ls.STREAMS and lag >= 0
        for n in names:
            assert n not in seen
            seen[n] = (st, lag, len(seen))
    assert sorted(seen) == sorted(ls.ACCESS)
    assert plan["period"] % 2 == 0, "the frame-counter words alternate by frame parity: the graphs' phases must keep it"
    # a launch sits behind its same-frame producers: same unit and later, or a larger lag; forward plans: never an earlier stream
    writer = {buf: n for n, (_, ws) in ls.ACCESS.items() for buf, off in ws}
    pos = {n: k for k, n in enumerate(itertools.chain.from_iterable(u[2] for u in plan["units"]))}
    for n, (reads, _) in ls.ACCESS.items():
        for buf, off in reads:
            if off == 0:
                w = writer[buf]
                assert seen[w][1] < seen[n][1] or (seen[w][:2] == seen[n][:2] and pos[w] < pos[n]), f"{n} before its producer {w}"
                if plan["forward"]:
                    assert seen[w][0] <= seen[n][0]
    if plan["forward"]:
        assert plan["period"] >= plan["group"] + ls.depth(plan) + 1


@pytest.mark.parametrize("name", sorted(ls.PLANS))
def test_captured_graphs_are_the_replays_modulo_the_period(name):
    """The engine captures each graph once on representative frame numbers; a replay is valid for the actual frames if the active launches,
    their ring slots (frame % period, and the previous frame's) and the frame parities agree."""
    plan = ls.PLANS[name]
    P = plan["period"]
    graphs = ls.graph_plans(plan)
    assert len(graphs) == 4 * P

    def shape(its):
        return [(e[0], e[1], e[2]) if e[0] == "wait" else (e[0], e[1], e[2], e[3], e[4] % P, (e[4] - 1) % P, e[4] % 2) for e in ls.events(plan, its)]
    for f, n in itertools.product(range(0, 2 * P + 1), (ls.depth(plan), 5, 6, 7, 11, 17, 30)):
        for key, its in ls.replays(plan, f, n):
            if key is not None:
                assert shape(its) == shape(graphs[key]), f"plan {name}: run({n}) from frame {f}, replay {key}"
        # a whole-run graph captured at frame f serves every later run of n frames from the same phase
        (key, its), = ls.replays(plan, f, n, whole=True)
        (key2, its2), = ls.replays(plan, f + 3 * P, n, whole=True)
        assert key == key2 == ("run", f % P, n) and shape(its) == shape(its2)


def test_the_model_finds_a_ring_that_is_too_short():
    """(the checker has teeth) plan F with rings of eight: stream 0 may run a whole replay ahead of stream 3 and lap it."""
    plan = dict(ls.PLANS["F"], period=8)
    m = check_runs(plan, (30,), True)
    assert any("overwrites" in e or "may already sit" in e for e in m.errors)
    plan = dict(ls.PLANS["B"], period=4)
    m = check_runs(plan, (30,), True)
    assert m.errors
    # ... and a forward plan without its waits
    plan = dict(ls.PLANS["F"])
    ev = ls.events(plan, [(i, 0, 30) for i in range(0, 33)])
    m = Model(plan)
    m.run([e for k, e in enumerate(ev) if e[0] == "launch" or k < ls.STREAMS])      # only the fork: no hand-over waits at all
    assert any("without waiting" in e for e in m.errors)


@pytest.mark.parametrize("name", sorted(ls.PLANS))
def test_engine_launches_touch_the_buffers_the_model_assumes(name):
    """RolloutEngine._lv_segments (the closures that make the ABI calls) against level_schedule.ACCESS: the engine's method runs on a stand-in
    whose buffers are labels and whose launch helpers record what they were handed -- every launch must read and write exactly the ring slots
    the happens-before model checked."""
    import types
    from urnn_amd.rollout import RolloutEngine
    plan = ls.PLANS[name]
    P = plan["period"]
    log = []
    ring = lambda b: [f"{b}[{k}]" for k in range(P)]

    def stage(label):
        return lambda inp, out=None: log.append((label, [inp], [out]))
    enc = types.SimpleNamespace(rnn1="enc1", rnn2="enc2", rnn3="enc3", stage2=stage("conv2"), stage3=stage("conv3"))
    dec = types.SimpleNamespace(rnn1="dec1", rnn2="dec2", rnn3="dec3", stage3=stage("deconv3"), stage2=stage("deconv2"))
    head = types.SimpleNamespace(run=lambda feat, **kw: log.append(("head", [feat, kw["partial0"]], [])))
    eng = types.SimpleNamespace(
        _lvP=P, _plan=plan["units"], net=types.SimpleNamespace(encoder=enc, decoder=dec, head=head),
        _ring_e=[ring("e1"), ring("e2"), ring("e3")], _ring_d1=ring("d1"), _ring_d2=ring("d2"),
        _ring={b: ring(b) for b in ("a2", "a3", "u3", "u2", "feat")}, _k1part=ring("k1part"), _ws=[f"ws{u}" for u in range(len(plan["units"]))],
        states=[None] * 5 + ["d3[0]"], a1="a1[0]", te2=[0, 1], t2=[0, 1], out_masked=None, out_cls=None, out_raw=None, _head_coop=False,
        _stem_stats=lambda: True,
        _stage1=lambda t, t_next=None: log.append(("stage1", [], ["a1[0]"])),
        _cell=lambda nm, cell, x, e, h, out, ws: log.append((nm, [v for v in (x, e, h) if v is not None], [out])),
        _last_conv=lambda d3, feat, k1: log.append(("lastconv", [d3], [feat, k1])))
    size = lambda b: P if ls.RINGS[b] == "period" else 1
    for u, (st, lag, names) in enumerate(plan["units"]):
        for tau in range(0, 2 * P + 3):
            del log[:]
            for seg in RolloutEngine._lv_segments(eng, u, tau):
                seg()
            assert [l[0] for l in log] == list(names)
            for nm, reads, writes in log:
                want_r = sorted(f"{b}[{(tau + off) % size(b)}]" for b, off in ls.ACCESS[nm][0])
                want_w = sorted(f"{b}[{(tau + off) % size(b)}]" for b, off in ls.ACCESS[nm][1])
                assert sorted(reads) == want_r and sorted(writes) == want_w, f"plan {name}: {nm}({tau}) reads {reads} writes {writes}"
