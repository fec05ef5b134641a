"""The level pipeline's schedule as data (no torch, no GPU): which launch runs on which stream in which iteration, which stream waits for
which, and which buffers a launch reads and writes.  ``rollout.RolloutEngine`` executes these event lists on HIP streams (eagerly or under
stream capture); ``tests/test_level_schedule.py`` replays the same lists through a happens-before model and proves that no launch can read
a ring slot before the frame it wants has been written there, or after a later frame has overwritten it -- whatever the timing.

A timestep of the network (model.py:65-121; encoder.py:119-215, decoder.py:102-217) as thirteen launches:

    stage1 -> enc1 -> conv2 -> enc2 -> conv3 -> enc3 -> dec3 -> deconv3 -> dec2 -> deconv2 -> dec1 -> lastconv -> head

Only a cell's own state recurs from frame to frame, so a PLAN cuts the chain into units, puts unit u on a stream and lets it work on frame
i - lag(u) in iteration i.  Buffers handed from one unit to a later one are rings of ``period`` slots indexed by frame % period.
"""

CUR = -1            # the stream the schedule forks from and joins into (the capturing stream); it never launches anything
STREAMS = 4         # the HIP runtime multiplexes streams onto four hardware queues

# (stream, lag, launches) per unit.  forward: every hand-over goes to a later stream; inside a replay stream q + 1 waits for streams 0 .. q
# and nothing waits for a later stream.  Otherwise all streams meet in a barrier between iterations.  period: slots per ring (even: the
# frame-counter words alternate by frame parity); group: steady iterations per replay.
PLANS = {
    "A": dict(forward=False, period=6, group=6, units=(
        (0, 0, ("stage1", "enc1", "conv2")), (1, 1, ("enc2", "conv3", "enc3")), (2, 2, ("dec3", "deconv3", "dec2")),
        (0, 3, ("deconv2",)), (3, 4, ("dec1", "lastconv", "head")))),
    "B": dict(forward=False, period=6, group=6, units=(
        (0, 0, ("stage1", "enc1", "conv2")), (1, 1, ("enc2", "conv3", "enc3")), (0, 2, ("dec3",)),
        (2, 3, ("deconv3", "dec2", "deconv2")), (3, 4, ("dec1", "lastconv", "head")))),
    "F": dict(forward=True, period=10, group=6, units=(
        (0, 0, ("stage1", "enc1", "conv2", "enc2")), (1, 1, ("conv3", "enc3", "dec3")), (2, 2, ("deconv3", "dec2", "deconv2")),
        (3, 3, ("dec1", "lastconv", "head")))),
}

# What a launch touches: (reads, writes) as (buffer, frame offset): ("e1", 0) = the slot of this frame, ("e1", -1) = the previous frame's.
# RINGS says how many slots a buffer has ("period" = the plan's; 1 = a single buffer, e.g. dec1's state d3, updated in place).  The
# unit's scratch and the frame-counter words are per stream and left out (one stream is one order), like the event's static inputs and
# the output rows (one row per frame, never reused).
ACCESS = {
    "stage1":   ((),                                              (("a1", 0),)),
    "enc1":     ((("a1", 0), ("e1", -1)),                         (("e1", 0),)),
    "conv2":    ((("e1", 0),),                                    (("a2", 0),)),
    "enc2":     ((("a2", 0), ("e2", -1)),                         (("e2", 0),)),
    "conv3":    ((("e2", 0),),                                    (("a3", 0),)),
    "enc3":     ((("a3", 0), ("e3", -1)),                         (("e3", 0),)),
    "dec3":     ((("e3", 0), ("d1", -1)),                         (("d1", 0),)),
    "deconv3":  ((("d1", 0),),                                    (("u3", 0),)),
    "dec2":     ((("u3", 0), ("e2", 0), ("d2", -1)),              (("d2", 0),)),
    "deconv2":  ((("d2", 0),),                                    (("u2", 0),)),
    "dec1":     ((("u2", 0), ("e1", 0), ("d3", -1)),              (("d3", 0),)),
    "lastconv": ((("d3", 0),),                                    (("feat", 0), ("k1part", 0))),
    "head":     ((("feat", 0), ("k1part", 0)),                    ()),
}
RINGS = {"a1": 1, "d3": 1, "e1": "period", "e2": "period", "e3": "period", "d1": "period", "d2": "period", "a2": "period", "a3": "period",
         "u3": "period", "u2": "period", "feat": "period", "k1part": "period"}


def depth(plan):
    return max(lag for _, lag, _ in plan["units"])


def events(plan, its, order=None):
    """The stream operations of ``its`` = [(i, lo, hi), ...] (iteration i over the frames lo <= t < hi), back to back in ONE replay:
    ("wait", dst, src) -- stream dst waits for everything stream src holds at this point -- and ("launch", stream, unit, name, frame).
    The side streams fork from CUR, and CUR joins them at the end.  Between iterations: a barrier through CUR (plans A / B, and every
    ``group`` iterations of a forward plan), else stream q + 1 waits for streams 0 .. q through CUR.  Within an iteration the units'
    launches are enqueued round-robin (a graph replay hands its nodes to the hardware queues in creation order)."""
    units = plan["units"]
    order = list(order) if order else list(range(len(units)))
    ev = [("wait", q, CUR) for q in range(STREAMS)]
    for n, (i, lo, hi) in enumerate(its):
        if n and (not plan["forward"] or n % plan["group"] == 0):
            # (side streams waiting for each other directly -- all-to-all, or only producer -> consumer -- end in a segmentation fault
            # inside hipStreamEndCapture on ROCm 7.0; through the capturing stream the same dependencies capture fine)
            ev += [("wait", CUR, q) for q in range(STREAMS)] + [("wait", q, CUR) for q in range(STREAMS)]
        elif n:
            for q in range(STREAMS - 1):
                ev += [("wait", CUR, q), ("wait", q + 1, CUR)]
        act = [(u, units[u][0], [(name, i - units[u][1]) for name in units[u][2]]) for u in order if lo <= i - units[u][1] < hi]
        for j in range(max(len(launches) for _, _, launches in act)):
            for u, st, launches in act:
                if j < len(launches):
                    ev.append(("launch", st, u, launches[j][0], launches[j][1]))
    ev += [("wait", CUR, q) for q in range(STREAMS)]
    return ev


def replays(plan, f, n, graphs=True, whole=False):
    """run(n) from frame f as the engine cuts it into replays: [(key, its), ...] -- fill, groups, single steady iterations, drain
    (key = (kind, frame phase), the captured graph that is replayed), or one eager pass when the run is shorter than the pipeline is deep.
    whole: the run as ONE replay, key ("run", frame phase, n) -- the graph the engine captures for a run length it has seen before (an
    event length): its barriers every ``group`` iterations cost less than the ends of eight replays."""
    P, D, G = plan["period"], depth(plan), plan["group"]
    end = f + n
    if not graphs or n < max(D, 1) or whole:
        return [(("run", f % P, n) if (graphs and whole) else None, [(i, f, end) for i in range(f, end + D)])]
    big = 1 << 30
    out = [(("fill", f % P), [(f + k, f, big) for k in range(D)])]
    i = f + D
    while i + G <= end:
        out.append((("group", i % P), [(i + k, 0, big) for k in range(G)]))
        i += G
    while i < end:
        out.append((("steady", i % P), [(i, 0, big)]))
        i += 1
    out.append((("drain", end % P), [(end + k, 0, end) for k in range(D)]))
    return out


def graph_plans(plan):
    """Every graph the engine captures: {(kind, phase): its}, on frame numbers away from zero (only frame % period and the parity matter)."""
    P, D, G = plan["period"], depth(plan), plan["group"]
    big = 1 << 30
    out = {}
    for p in range(P):
        f = 2 * P + p
        out[("fill", p)] = [(f + k, f, big) for k in range(D)]
        out[("steady", p)] = [(f, 0, big)]
        out[("group", p)] = [(f + k, 0, big) for k in range(G)]
        out[("drain", p)] = [(f + k, 0, f) for k in range(D)]
    return out
