"""On-device rollout engine: the T-step loop of ``test.Inference`` (test.py:326-377) with everything resident
in HBM -- six hidden states updated in place, per-frame input assembly, the whole timestep (about 45 kernel
launches) captured once as a hipGraph and replayed T times while a device-side frame counter advances.
No host round trip inside the loop; one D2H of the (T,B,H,W) result at the end if the caller wants numpy."""
import os

import torch

from . import level_schedule, ops
from .ops import tuning_env as _tuning_env
from ._lib import lib
from .dataset import event_to_device
from .general import initialize_states


class RolloutEngine:
    def __init__(self, net, input_height, input_width, historical_nums, rain_max, cumsum_rain_max, batch=1,
                 max_frames=360, spatial_rain=False, net_cfg=None, use_graph=True, keep_raw=False,
                 device=None, overlap=False, fused_reset_gate=True, coop_cells=True, fused_tails=False, coop_mutex=False, levels=None):
        self.net = net
        # cells whose shape has the form recompute the reset gate inside the candidate kernel instead of round-tripping its raw
        # planes through HBM (include/urnn_hip.h URNN_PHASE_FUSED_R); the engine never reads a cell's workspace
        # ... and cells on small planes run as ONE cooperative launch (URNN_PHASE_COOP: gates | grid barrier | candidate | grid
        # barrier | blend, nothing but statistics leaving the CU in between): per cell, self._coop below
        self._cell_flags = ops.PHASE_FUSED_R if fused_reset_gate else 0
        self.H, self.W = int(input_height), int(input_width)
        self.nums = int(historical_nums)
        self.C = 2 * self.nums + 3
        self.rain_max, self.cumsum_max = float(rain_max), float(cumsum_rain_max)
        self.B = int(batch)
        self.Tcap = int(max_frames)
        self.spatial = bool(spatial_rain)
        self.use_graph = bool(use_graph)
        # overlap=True: head(t-1), encoder(t+1) and decoder(t) run as concurrent kernel chains (streams forked inside the
        # captured graph: three, or two with the head in front of the encoder -- _head_own_chain); needs ping-pong encoder
        # states and feature maps.  Same arithmetic, same results.
        self.overlap = bool(overlap)
        # levels=True (default on small planes with overlap=True): the LEVEL PIPELINE -- four concurrent kernel chains cut by level of the
        # network, each a frame behind the one that feeds it (see "level pipeline" below).  Same launches, same results.
        if levels is None:
            px = int(batch) * int(input_height) * int(input_width)
            levels = self.overlap and not fused_tails and not coop_mutex and (px <= self.LEVELS_MAX_PIXELS or (int(batch) == 1 and px <= self.LEVELS_MAX_PIXELS_ONE))
            if _tuning_env("URNN_TUNE_LEVELS", "") in ("0", "1"):
                levels = self.overlap and _tuning_env("URNN_TUNE_LEVELS", "") == "1"
        self.levels = bool(levels) and self.overlap
        self.device = torch.device(device) if device is not None else next(net.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("RolloutEngine needs the model on a GPU (HIP) device")
        self.net_cfg = net_cfg
        B, H, W, dev = self.B, self.H, self.W, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        # static event buffers (filled per event by load_event)
        # rows of the per-frame buffers: the capture warm-up of the overlapped schedules runs a few frames for real (two; the level pipeline eight)
        self._rows = max(self.Tcap, 8 if self.levels else 2)
        rshape = (B, self._rows + 1, H, W) if self.spatial else (B, self._rows + 1)
        self.rain = torch.zeros(rshape, **f32)
        self.cumsum = torch.zeros(rshape, **f32)
        self.dem = torch.zeros((B, H, W), **f32)
        self.imperv = torch.zeros((B, H, W), **f32)
        self.manhole = torch.zeros((B, H, W), **f32)
        self.dem_min, self.dem_max = 0.0, 1.0
        # recurrent state, updated in place
        self.states = list(initialize_states(dev, H, W, net_cfg, batch=B))
        self.enc_alt = [torch.zeros_like(s) for s in self.states[:3]] if (self.overlap and not self.levels) else None
        self._frames_done = 0
        # activations between kernels
        enc, dec = net.encoder, net.decoder
        self.x_in = torch.empty((B, self.C, H, W), **f32)
        self.a1 = torch.empty((B, enc.stage1.out_channels, H, W), **f32)
        # scalar rain: stage 1 collapses to LeakyReLU(S + v_t); S is refreshed by load_event
        self.S1 = None if self.spatial else torch.zeros((B, enc.stage1.out_channels, H, W), **f32)
        c1 = enc.stage1.layer
        self._w1 = c1.weight.detach().reshape(c1.out_channels, -1).clone()    # pointer-stable copy (captured in the graph)
        self.a2 = torch.zeros((B, enc.stage2.out_channels, H // 2, W // 2), **f32)
        self.a3 = torch.zeros((B, enc.stage3.out_channels, H // 4, W // 4), **f32)
        self.u3 = torch.zeros((B, dec.stage3.out_channels, H // 2, W // 2), **f32)
        self.u2 = torch.zeros((B, dec.stage2.out_channels, H, W), **f32)
        self.feat = torch.zeros((B, dec.stage1.out_channels, H, W), **f32)
        self.feat_alt = torch.empty_like(self.feat) if self.overlap else None   # overlap mode: head(t-1) || decoder(t)
        # outputs for every frame
        rows = self._rows
        self.out_masked = torch.zeros((rows, B, H, W), **f32)
        self.out_cls = torch.zeros((rows, B, H, W), **f32)
        self.out_raw = torch.zeros((rows, B, H, W), **f32) if keep_raw else None
        # the frame counters share one buffer (reset() zeroes it in one launch)
        self._counters = torch.zeros(8, dtype=torch.int32, device=dev)
        self.t_dev = self._counters[0:1]                              # one chain: the frame index (urnn_advance_counter per frame)
        # overlap mode: a pair of words per kernel family, used alternately by frame parity -- the head of frame t reads t2[t % 2] and
        # stores t + 1 to t2[1 - t % 2], the input assembly of frame t likewise with te2 (urnn_*_rollout_f32): no counter kernels
        self.t2 = self._counters[2:4]
        self.te2 = self._counters[4:6]
        self.zero_frame = torch.zeros(1, dtype=torch.int32, device=dev)
        # scratch: OWNED by this engine (a captured graph holds raw pointers into it), one buffer per concurrent kernel chain,
        # sized for the largest consumer before any capture
        L = lib()
        need = max([L.urnn_head_workspace_bytes(B, 16, H, W)] +
                   [L.urnn_gru_cell_workspace_bytes(B, c.num_features, c.shape[0], c.shape[1])
                    for c in (enc.rnn1, enc.rnn2, enc.rnn3, dec.rnn3, dec.rnn2, dec.rnn1)])
        # (three chains: the third is the head's; level pipeline: one per unit.)  One allocation, so that check_status() reads every status
        # word in ONE transfer -- five device reads per event were 4 % of a 30-frame event at 64x64
        nws = 5 if self.levels else 3 if self.overlap else 1
        self._ws_stride = (int(need) + 255) // 256 * 256
        self._ws_all = ops.workspace(self._ws_stride * nws, dev)
        self._ws_all.view(nws, self._ws_stride)[:, :ops.STATUS_AREA_BYTES].zero_()
        self._ws = [self._ws_all[k * self._ws_stride:k * self._ws_stride + int(need)] for k in range(nws)]
        # fused_tails=True: the END of a cell runs together with the stage conv that consumes the new state (ops.gru_cell_tail: blend +
        # 1x1 conv [+ pool], for the decoder's last cell + the head's first LayerNorm statistics): enc1 -> stage2, enc2 -> stage3,
        # dec1 -> stage1.  190 MB per frame less through HBM at 500x500, identical bits -- and 3-4 % FEWER frames/s (DESIGN.md section
        # 4.10: the fused kernel reads 256-byte runs per plane where the blend reads 4 KB ones), hence off by default
        self._fused_tails = bool(fused_tails)
        self._tails = None          # decided at the first step (the layers' weight ranges are known once they are packed)
        self._stem = None           # likewise: the decoder's last conv takes the head's first statistics (_stem_stats)
        self._k1part = [ops.head_tail_partial(B, H, W, dev) for _ in range(2 if self.overlap else 1)]      # (level pipeline: a ring, below)
        self._param_stamp = None    # what the captured graphs' packed weights were made from (see _check_params)
        self._params = None
        self._run_seen = {}         # overlapped schedules: how often run(n) came from a frame phase (see _run_overlap / _run_levels)
        self._probe = None          # {"enc1": [(start, stop), ...], "dec1": [...]} while probing
        self._graph = None
        self._graph_group = None
        self._graphs2 = None
        # (equal stream priorities: a high-priority chain starves the other -- 900 instead of 1 250 frames/s either way round)
        self._side = tuple(torch.cuda.Stream(device=dev) for _ in range(self.LEVEL_STREAMS if self.levels else 3)) if self.overlap else None
        # A cooperative cell launch needs ALL its blocks resident (one per CU).  With the encoder and decoder chains in flight, two such
        # launches of at most 128 blocks each always fit side by side (the head's: further below); a larger one could wait at its grid barrier for CUs the other chain's
        # cooperative launch holds while that one waits for CUs of ours.  So with overlap=True only cells of <= 128 blocks take the
        # flag (measured at 500x500, 245 blocks: ordering the two chains' launches instead costs more than the launch saves,
        # profiles/r04_coop_cells.txt); one chain takes it wherever the library plans it.
        resident = {}                                                # blocks of the cooperative launches in use, by layer
        cus = int(torch.cuda.get_device_properties(self.device).multi_processor_count)   # 256 on an MI355X (the library bounds its plans by the same count)

        # Round 6, measured and left OFF (coop_mutex=True / URNN_TUNING=1 URNN_TUNE_COOP_MUTEX=1 turns it on): a cooperative launch of MORE
        # than half the chip's CUs (the quarter-resolution cells' 245 blocks, the half-resolution cells' 245 persistent blocks of four
        # tiles, urnn_coop_tiles.hip) can run next to other chains as long as no two of them are in flight at once -- the engine orders
        # them with one event edge per iteration (``_big_next``; kernels without a grid barrier always finish and free their CUs, so a
        # lone cooperative launch only ever waits).  Correct (the full-size rollout tests pass on it), and 8.5 % SLOWER than the three
        # kernels per cell: 1 354 against 1 480 frames/s on one box (profiles/r06_ab_coop_mutex.txt).  A launch that needs every CU
        # drains the other two chains before it can start and idles the chip while its blocks trickle in -- the overlapped schedule
        # lives on kernels that share CUs.  The cooperative half-resolution cells therefore serve the ONE-chain schedule
        # (overlap=False, urnn_step_f32: +6 %), the three-chain schedule keeps the three kernels.
        self._coop_mutex = self.overlap and bool(coop_cells) and (coop_mutex or _tuning_env("URNN_TUNE_COOP_MUTEX", "0") != "0")
        self._mutex_small_only = _tuning_env("URNN_TUNE_COOP_MUTEX", "0") == "2"      # (A/B: only the one-tile-per-block cells join the order)
        self._mutex_tiles_only = _tuning_env("URNN_TUNE_COOP_MUTEX", "0") == "3"      # (A/B: only the four-tiles-per-block cells)
        self._big_coop = {}
        self._big_next = {}
        names = iter(("enc1", "enc2", "enc3", "dec3", "dec2", "dec1"))

        def coop_flag(cell, has_x, skip):
            n = L.urnn_gru_cell_coop_blocks(B, cell.input_channels, cell.num_features, cell.shape[0], cell.shape[1], int(skip), int(has_x))
            # (URNN_TUNE_COOP_BIG=0: a one-chain engine keeps the two-chain policy -- the counter passes of tools/collect_profiles.sh run
            # eager on one chain and must execute the kernels of the benchmarked schedule)
            big_ok = (not self.overlap and _tuning_env("URNN_TUNE_COOP_BIG", "1") != "0") or self._coop_mutex
            # more 64-pixel tiles than CUs: the launch is the four-tiles-per-block one (urnn_coop_tiles.hip), whose blocks take a whole CU each
            # (159 KB of LDS, every register): next to other chains it waits for CUs to drain whatever its block count -- one chain only
            whole_cu = B * ((cell.shape[0] * cell.shape[1] + 63) // 64) > cus
            use = coop_cells and n > 0 and (n <= cus // 2 or big_ok) and (not whole_cu or not self.overlap or (self._coop_mutex and not self._mutex_small_only))
            if self.overlap and self._mutex_tiles_only and not whole_cu and n > cus // 2:
                use = False
            self._big_coop[next(names)] = bool(use and self.overlap and (n > cus // 2 or whole_cu))
            resident[len(resident)] = n if (use and n <= cus // 2) else 0          # (the large ones are serialised: they never add up)
            return ops.PHASE_COOP if use else 0
        nhead = L.urnn_head_coop_blocks_f32(B, H, W)                 # ... and the head likewise (urnn_head_coop_f32)
        self._head_coop = bool(coop_cells) and nhead > 0 and (nhead <= cus // 2 or (not self.overlap and nhead <= cus and _tuning_env("URNN_TUNE_COOP_BIG", "1") != "0"))
        self._coop = {"enc1": coop_flag(enc.rnn1, 1, 0), "enc2": coop_flag(enc.rnn2, 1, 0), "enc3": coop_flag(enc.rnn3, 1, 0),
                      "dec3": coop_flag(dec.rnn3, 0, 1), "dec2": coop_flag(dec.rnn2, 1, 1), "dec1": coop_flag(dec.rnn1, 1, 1)}
        # The head as a THIRD chain (own stream and scratch): head(t-1) || encoder(t+1) || decoder(t) -- +2 % at 500x500, +4 % at 400x560
        # (profiles/r04_three_chains.txt).  With cooperative launches on three streams the residency rule above becomes: the largest
        # such launch of each chain, together, must fit the chip's 256 CUs (a block each) -- else the head stays in front of the encoder.
        blocks = list(resident.values())                             # enc1..3, dec3..1 in the order of the dict above
        together = max(blocks[:3]) + max(blocks[3:]) + (nhead if self._head_coop else 0)
        self._head_own_chain = self.overlap and _tuning_env("URNN_TUNE_HEAD_CHAIN", "1") != "0" and together <= cus
        if self.levels:
            # Level pipeline: four streams are in flight at once, so the largest cooperative launch of each, together, must fit the chip (a
            # block per CU); the largest ones give the flag back until they do.
            nb = dict(zip(("enc1", "enc2", "enc3", "dec3", "dec2", "dec1"), blocks))
            nb["head"] = nhead if self._head_coop else 0
            # (measured, iteration time in us, A / B / F: 64x64 91 / 83 / 79-84, 52x120 93 / 97 / 88, 128x128 110 / 116 / 104; whole events with
            # the event-length graph, frames/s B | F: 64x64 11 505 | 11 721, 52x120 10 051 | 10 942, 128x128 8 528 | 9 143 -- r06_level_pipeline.txt)
            pname = _tuning_env("URNN_TUNE_LEVEL_PLAN", "F")
            if pname == "AB":
                pname = "B" if (self._coop["enc1"] and nb["enc1"] <= 64) else "A"
            plan = dict(self.LEVEL_PLANS[pname])
            if plan["forward"] and _tuning_env("URNN_TUNE_LEVEL_GROUP", ""):                     # (A/B of the replay length: rings follow)
                plan["group"] = int(_tuning_env("URNN_TUNE_LEVEL_GROUP", ""))
                plan["period"] = (plan["group"] + level_schedule.depth(plan) + 2) & ~1
            self._plan_dict, self._plan, self._lvP = plan, plan["units"], plan["period"]
            per_stream = [[n for st, _, names_ in self._plan if st == q for n in names_ if n in nb] for q in range(self.LEVEL_STREAMS)]   # (one launch per stream at a time)
            while sum(max([nb[n] for n in names_] or [0]) for names_ in per_stream) > cus:
                big = max(nb, key=nb.get)
                nb[big] = 0
                if big == "head":
                    self._head_coop = False
                else:
                    self._coop[big] = 0
            self._head_own_chain = True
            P = self._lvP
            self._k1part = [ops.head_tail_partial(B, H, W, dev) for _ in range(P)]
            self._ring_e = [[torch.zeros_like(st) for _ in range(P)] for st in self.states[:3]]
            self._ring_d1 = [torch.zeros_like(self.states[3]) for _ in range(P)]
            self._ring_d2 = [torch.zeros_like(self.states[4]) for _ in range(P)]
            self._ring = {k: [t] + [torch.zeros_like(t) for _ in range(P - 1)]      # (zeros: the capture warm-up reads slots no frame has written yet)
                          for k, t in (("a2", self.a2), ("a3", self.a3), ("u3", self.u3), ("u2", self.u2), ("feat", self.feat))}
        self._dem_stamp = None
        # What an event starts from -- the six zero states (general.py:50-95; the overlapped schedules: both copies of the encoder's; the level
        # pipeline: the ring slot the first frame reads as "previous", period - 1) -- lives in ONE buffer, so that reset() is two launches
        # (states, counters) instead of one per tensor: 59 fills of ~4 us of host time each were 5 % of a 30-frame event at 64x64.
        homes = [(self.states, k) for k in range(6)] + ([(self.enc_alt, k) for k in range(3)] if self.enc_alt is not None else [])
        if self.levels:
            homes = [(ring, self._lvP - 1) for ring in self._ring_e + [self._ring_d1, self._ring_d2]] + [(self.states, 5)]
        sizes = [(lst[k].numel() + 63) // 64 * 64 for lst, k in homes]                     # (every view 256-byte aligned)
        self._zeros = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        off = 0
        for (lst, k), n in zip(homes, sizes):
            lst[k] = self._zeros[off:off + lst[k].numel()].view(lst[k].shape)
            off += n

    # -- one timestep, all launches on the current stream ----------------------------------------------
    def _step(self):
        net = self.net
        enc, dec = net.encoder, net.decoder
        e1, e2, e3, d1, d2, d3 = self.states
        ws = self._ws[0]
        self._stage1(self.t_dev)
        if not self._cell("enc1", enc.rnn1, self.a1, None, e1, e1, ws, conv_out=self.a2):
            enc.stage2(e1, out=self.a2)
        if not self._cell("enc2", enc.rnn2, self.a2, None, e2, e2, ws, conv_out=self.a3):
            enc.stage3(e2, out=self.a3)
        self._cell("enc3", enc.rnn3, self.a3, None, e3, e3, ws)
        self._cell("dec3", dec.rnn3, None, e3, d1, d1, ws)
        dec.stage3(d1, out=self.u3)
        self._cell("dec2", dec.rnn2, self.u3, e2, d2, d2, ws)
        dec.stage2(d2, out=self.u2)
        if not self._cell("dec1", dec.rnn1, self.u2, e1, d3, d3, ws, conv_out=self.feat, k1part=self._k1part[0]):
            self._last_conv(d3, self.feat, self._k1part[0])
        tail = self._tail_of("dec1") is not None or self._stem_stats()
        net.head.run(self.feat, out_masked=self.out_masked, out_cls=self.out_cls, out_raw=self.out_raw,
                     frame_index=self.t_dev, ws=ws, partial0=self._k1part[0] if tail else None, coop=self._head_coop and not tail)
        ops.advance_counter(self.t_dev, 1)

    def _stage1(self, t_dev, t_next=None):
        """Frame input assembly + encoder stage-1 conv -> self.a1.  ``t_next``: the word that receives t_dev + 1 (ops.preprocess)."""
        conv = self.net.encoder.stage1.layer
        if self.S1 is not None:
            ops.stage1_scalar_rain(self.S1, self.rain, self.cumsum, self._w1, conv.bias.detach(), 0, self.nums, self.rain_max,
                                   self.cumsum_max, out=self.a1, t_dev=t_dev, t_next=t_next)
        else:
            ops.preprocess(self.rain, self.cumsum, self.dem, self.imperv, self.manhole, self.dem_min, self.dem_max, 0,
                           self.nums, self.rain_max, self.cumsum_max, out=self.x_in, t_dev=t_dev, t_next=t_next)
            self.net.encoder.stage1(self.x_in, out=self.a1)

    def _stem_stats(self):
        """Does the decoder's last conv take the head's first LayerNorm statistics in its epilogue (ops.stage_conv_stem_applies: the big
        planes)?  Decided once per set of weights; never together with the fused cell tail or with a layer on the exact instruction."""
        if self._stem is None:
            st = self.net.decoder.stage1
            st._packed()                                              # (sets the layer's `wide` flag)
            self._stem = (self._tail_of("dec1") is None and not st._cache.wide and not self._head_coop and
                          ops.stage_conv_stem_applies(self.B, st.layer.in_channels, st.out_channels, self.H, self.W))
        return self._stem

    def _last_conv(self, d3, feat, k1part):
        """Decoder.stage1 (64 -> 16 + LeakyReLU) into ``feat``; with ``_stem_stats`` also the head's first pass (its partials -> k1part)."""
        if self._stem_stats():
            self.net.decoder.stage1(d3, out=feat, head_w=self.net.head.flat_params()["conv_w"], head_partial0=k1part)
        else:
            self.net.decoder.stage1(d3, out=feat)

    def _tail_of(self, name):
        """The stage conv fused into the named cell's last kernel, or None (decided once per set of weights)."""
        if self._tails is None:
            enc, dec = self.net.encoder, self.net.decoder
            self._tails = {}
            for cname, cell, stage in (("enc1", enc.rnn1, enc.stage2), ("enc2", enc.rnn2, enc.stage3), ("dec1", dec.rnn1, dec.stage1)):
                cell._packed(), stage._packed()                      # (sets the layers' `wide` flags)
                ok = (self._fused_tails and stage.kind == "conv" and not self._coop[cname] and not cell._cache.wide and not stage._cache.wide and
                      ops.gru_cell_tail_applies(self.B, cell.num_features, cell.shape[0], cell.shape[1], stage.out_channels, stage.pool))
                self._tails[cname] = stage if ok else None
        return self._tails.get(name)

    def _cell(self, name, cell, x, e, h, out, ws, conv_out=None, k1part=None):
        """One GRU cell (+ the stage conv behind it when ``conv_out`` is given: fused into the cell's last kernel where that form
        exists, its own launch otherwise).  When a probe is active the three kernels of the named cells (gate GEMM | candidate GEMM |
        GroupNorm finalize + blend [+ conv]) are launched one ABI call each, every one bracketed by events on the launch stream
        (bench.py's live roofline measurement)."""
        stage = self._tail_of(name) if conv_out is not None else None
        head_w = self.net.head.flat_params()["conv_w"][0] if (stage is not None and k1part is not None) else None
        flags = self._cell_flags | self._coop[name]
        # a cooperative launch of more than half the chip inside an overlapped iteration: the NEXT such launch, if another chain's, waits for it
        notify = self._big_next.get(name) if self._probe is None else None

        def run(mask):
            if stage is not None and (mask & ops.PHASE_BLEND):
                cell.step_tail(x, e, h, stage, conv_out, out=out, phases=mask | self._cell_flags, ws=ws, head_w=head_w,
                               head_partial0=k1part if head_w is not None else None)
            else:
                cell.step(x, e, h, out=out, phases=mask | (flags if mask == ops.PHASE_ALL else self._cell_flags), ws=ws)
        if self._probe is not None and name in self._probe:
            for kind, mask in (("gates", ops.PHASE_GATES), ("candidate", ops.PHASE_GN1 | ops.PHASE_CAND), ("blend", ops.PHASE_GN2 | ops.PHASE_BLEND)):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                run(mask)
                b.record()
                self._probe[name][kind].append((a, b))
        else:
            run(ops.PHASE_ALL)
        if notify is not None:
            # The other chain's stream waits IMMEDIATELY (its later launches queue behind the wait; what it holds already is unaffected): a
            # wait captured after the recording stream had moved on, and a record nobody waits for, both crashed hipStreamEndCapture (ROCm 7.0).
            evt = torch.cuda.Event()
            evt.record(torch.cuda.current_stream(self.device))
            notify.wait_event(evt)
        return stage is not None          # True: conv_out has been written

    PROBED_CELLS = ("enc1", "dec1", "enc2", "dec2")

    def probe_cell_kernels(self, frames=12):
        """Average duration (seconds) of the gate-GEMM / candidate-GEMM / blend launches of the full- and half-resolution cells
        while the rollout runs in this engine's scheduling mode (eager launches, same streams and co-running kernels as the
        captured graph): {cell: {"gates": s, "candidate": s, "blend": s}}.  An event pair on the launch stream spans the kernel
        plus whatever it waited for on that stream's queue -- with overlap=False that is the kernel itself (+ ~1 us)."""
        saved_graph, self.use_graph = self.use_graph, False
        self.reset()
        self.run(2)
        self._probe = {c: {"gates": [], "candidate": [], "blend": []} for c in self.PROBED_CELLS}
        self.run(frames)
        torch.cuda.synchronize(self.device)
        probe, self._probe = self._probe, None
        self.use_graph = saved_graph
        self.reset()
        # (An event pair around one launch reads ~2 us more than the kernel's own begin-to-end time -- the record path and the launch
        # boundary: bench.py's figures sit 4-6 % above rocprofv3's for the same launches, profiles/r04_kernel_stats_overlap0.txt.  An
        # empty pair measures 6-8 us here and over-corrects; nothing is subtracted.)
        return {c: {k: sum(a.elapsed_time(b) for a, b in v) / len(v) / 1e3 for k, v in kinds.items()} for c, kinds in probe.items()}

    def probe_gate_gemm(self, frames=12):
        """(kept for callers of the round-2 interface) gate-GEMM durations only: {cell: seconds}"""
        return {c: v["gates"] for c, v in self.probe_cell_kernels(frames).items()}

    # -- overlap mode: two concurrent chains -------------------------------------------------------------
    def _enc_bufs(self, parity):
        """(read, write) encoder-state triples of E(t) with t % 2 == parity: writes e[parity], reads e[1 - parity]."""
        bufs = (self.states[:3], self.enc_alt)
        return bufs[1 - parity], bufs[parity]

    # The encoder and decoder chains as lists of segments (closures), so that an overlapped iteration can interleave their ENQUEUE order:
    # a graph replay hands its kernel nodes to the hardware queues in creation order at a few microseconds each, so a chain
    # enqueued entirely after the other starts ~100 us late.
    def _enc_segments(self, parity):
        """encoder pass of a frame of ``parity`` into the buffers of that parity (frame index: te2[parity], see __init__)."""
        enc = self.net.encoder
        (p1, p2, p3), (n1, n2, n3) = self._enc_bufs(parity)

        ws = self._ws[0]                       # chain 1 (head + encoder) scratch

        def e1():
            self._stage1(self.te2[parity:parity + 1], t_next=self.te2[1 - parity:2 - parity])
            if not self._cell("enc1", enc.rnn1, self.a1, None, p1, n1, ws, conv_out=self.a2):
                enc.stage2(n1, out=self.a2)

        def e2():
            if not self._cell("enc2", enc.rnn2, self.a2, None, p2, n2, ws, conv_out=self.a3):
                enc.stage3(n2, out=self.a3)

        def e3():
            self._cell("enc3", enc.rnn3, self.a3, None, p3, n3, ws)
        return [e1, e2, e3]

    def _dec_segments(self, parity):
        """decoder(t) for t % 2 == parity: reads encoder states e[parity], writes feat[parity]."""
        dec = self.net.decoder
        e1, e2, e3 = self._enc_bufs(parity)[1]   # encoder states of frame t (written by E(t))
        _, _, _, d1, d2, d3 = self.states

        ws = self._ws[1]                       # chain 2 (decoder) scratch

        def s3():
            self._cell("dec3", dec.rnn3, None, e3, d1, d1, ws)
            dec.stage3(d1, out=self.u3)

        def s2():
            self._cell("dec2", dec.rnn2, self.u3, e2, d2, d2, ws)
            dec.stage2(d2, out=self.u2)

        def s1():
            feat = self.feat if parity == 0 else self.feat_alt
            if not self._cell("dec1", dec.rnn1, self.u2, e1, d3, d3, ws, conv_out=feat, k1part=self._k1part[parity]):
                self._last_conv(d3, feat, self._k1part[parity])
        return [s3, s2, s1]

    def _enc_chain(self, parity):
        for seg in self._enc_segments(parity):
            seg()

    def _dec_chain(self, parity):
        for seg in self._dec_segments(parity):
            seg()

    def _head_chain(self, parity):
        """head(t) for t % 2 == parity (reads feat[parity], frame index t2[parity]).  On a chain of its own (stream, scratch) when
        ``_head_own_chain``, else in front of the encoder pass on chain 1."""
        tail = self._tail_of("dec1") is not None or self._stem_stats()
        self.net.head.run(self.feat if parity == 0 else self.feat_alt, out_masked=self.out_masked, out_cls=self.out_cls,
                          out_raw=self.out_raw, frame_index=self.t2[parity:parity + 1], ws=self._ws[2 if self._head_own_chain else 0],
                          partial0=self._k1part[parity] if tail else None, coop=self._head_coop and not tail,
                          frame_next=self.t2[1 - parity:2 - parity])

    def _iter_overlap(self, parity, with_head=True):
        """Iteration t (parity = t % 2): chain 1 = head(t-1) then encoder(t+1); chain 2 = decoder(t); enqueued segment by
        segment in the order ENQUEUE_ORDER (H head, E encoder segment, D decoder segment)."""
        cur = torch.cuda.current_stream(self.device)
        s1, s2 = self._side[:2]
        # mutual exclusion of the large cooperative launches (``_big_coop``): in enqueue order, each one followed by one of ANOTHER chain
        # makes that chain's stream wait for it (same chain: stream order does it)
        seq, ie_, id_ = [], 0, 0
        for ch in _tuning_env("URNN_TUNE_CHAIN_ORDER", self._enqueue_order()):
            if ch == "E":
                seq.append((("enc1", "enc2", "enc3")[ie_], s1))
                ie_ += 1
            elif ch == "D":
                seq.append((("dec3", "dec2", "dec1")[id_], s2))
                id_ += 1
        bigs = [(n, st) for n, st in seq if self._big_coop.get(n)]
        self._big_next = {n: (bigs[i + 1][1] if i + 1 < len(bigs) and bigs[i + 1][1] is not st else None) for i, (n, st) in enumerate(bigs)}
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        s3 = self._side[2] if self._head_own_chain else s1
        if with_head and s3 is not s1:
            s3.wait_stream(cur)
        enc = self._enc_segments(1 - parity)
        dec = self._dec_segments(parity)
        order = _tuning_env("URNN_TUNE_CHAIN_ORDER", self._enqueue_order())
        if sorted(order) != sorted("HEEEDDD"):
            raise RuntimeError("enqueue order must hold one H, three E and three D")
        ie = idd = 0
        for ch in order:
            if ch == "H":
                if with_head:
                    with torch.cuda.stream(s3):
                        self._head_chain(1 - parity)
            elif ch == "E":
                with torch.cuda.stream(s1):
                    enc[ie]()
                ie += 1
            elif ch == "D":
                with torch.cuda.stream(s2):
                    dec[idd]()
                idd += 1
        if ie != 3 or idd != 3:
            raise RuntimeError("enqueue order must hold one H, three E and three D")
        cur.wait_stream(s1)
        cur.wait_stream(s2)
        if with_head and s3 is not s1:
            cur.wait_stream(s3)
        self._big_next = {}

    # The decoder chain is the longest: its first segment goes first, then the chains alternate (profiles/r04_enqueue_order.txt:
    # against head-then-encoder-then-decoder +2.5 % at 64x64, +4.5 % at 128x128, +1 % at 52x120, +0.5..1 % at 500x500 / 400x560)
    # Round 6 (profiles/r06_ab_enqueue_order.txt, same-box alternations): on the big planes the head first and the decoder's two deep segments
    # before the encoder's first -- "HDDEDEE" -- is +0.9 % at 500x500 (1 490 -> 1 504, 1 477 -> 1 490 on two boxes) and +0.3 % at 400x560; on the
    # small grids it loses 2-3.5 % (64x64 6 741 -> 6 536, 128x128 5 753 -> 5 631): the order follows the plane size.
    ENQUEUE_ORDER = "DEHDEDE"
    ENQUEUE_ORDER_LARGE = "HDDEDEE"

    def _enqueue_order(self):
        return self.ENQUEUE_ORDER_LARGE if self.H * self.W >= 100000 else self.ENQUEUE_ORDER
    GROUP = 4

    def _capture_overlap(self):
        self.net.head.flat_params()
        saved = [s.clone() for s in self.states] + [s.clone() for s in self.enc_alt]
        t0, t1 = self.t2.clone(), self.te2.clone()
        # Warm-up and first replays run a few frames for real.  They start at an EVEN frame b (the graphs are per frame parity and
        # "prologue" / "first 0" are an even frame's), as close behind the frames done as the buffers allow; the output rows they
        # write are saved and restored (after a mid-event re-capture near the end of an event they may be finished frames).
        self._group = max(0, int(_tuning_env("URNN_TUNE_GROUP", self.GROUP))) & ~1
        rows = self.out_masked.shape[0]
        b = max(0, min(self._frames_done + (self._frames_done & 1), (rows - self._group - 2) & ~1, (rows - 2) & ~1))
        outs = [o for o in (self.out_masked, self.out_cls, self.out_raw) if o is not None]
        kept = [o[b:b + self._group + 2].clone() for o in outs]

        def set_counters():
            self.t2.copy_(torch.tensor([b, b + 1], dtype=torch.int32))
            self.te2.copy_(torch.tensor([b, b + 1], dtype=torch.int32))
        set_counters()
        self._enc_chain(0)              # warm-up (packs weights), eager
        self._iter_overlap(0, with_head=False)
        self._iter_overlap(1)
        self._head_chain(1)
        torch.cuda.synchronize(self.device)
        graphs = {}
        # steady-state iterations (head(t-1) + encoder(t+1) || decoder(t)) per frame parity, the first iteration of a run() call
        # (no head pending) per parity, and the trailing head that completes a run() call: a short run (bench.py --steps 20
        # after an odd warm-up) is then a handful of graph replays too instead of ~30 eager launches
        for key, parity, with_head in ((("first", 0), 0, False), (("first", 1), 1, False), (0, 0, True), (1, 1, True)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._iter_overlap(parity, with_head=with_head)
            graphs[key] = g
        # GROUP steady-state iterations per replay (an even number: parity p, 1 - p, ...): a graph launch costs the frame loop ~5 us of
        # idle GPU, which is 3 % of a 64x64 frame (profiles/r04_bench_configs.txt)
        if self._group:
            for parity in (0, 1):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    for k in range(self._group):
                        self._iter_overlap((parity + k) % 2)
                graphs[("group", parity)] = g
        for parity in (0, 1):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._head_chain(parity)
            graphs[("tail", parity)] = g
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._enc_chain(0)          # pipeline prologue E(0) of an event
        graphs["prologue"] = g
        # The first launch of an instantiated graph uploads it to the device: pay that here, not in the first frames of the first
        # event.  Short valid rollouts touch every graph and write frames b .. b + GROUP + 1 only (states, counters, outputs restored below).
        seqs = [(3, ("prologue", ("first", 0), 1, 0, ("tail", 0))), (2, ("prologue", ("first", 0), ("tail", 0), ("first", 1), ("tail", 1)))]
        if self._group:
            seqs += [(self._group + 1, ("prologue", ("first", 0), ("group", 1), ("tail", 0))),
                     (self._group + 2, ("prologue", ("first", 0), 1, ("group", 0), ("tail", 1)))]
        for nframes, seq in seqs:
            if b + nframes <= rows:
                set_counters()
                for key in seq:
                    graphs[key].replay()
                    # (one graph at a time here: a freshly instantiated graph uploads itself on its first launch, and a host-side
                    # segmentation fault inside hipGraphLaunch was seen once in ~10 runs of the GPU suite at exactly this replay,
                    # with several first launches in flight.  Untimed warm-up: the synchronisations cost nothing that is measured.)
                    torch.cuda.synchronize(self.device)
        torch.cuda.synchronize(self.device)
        self._graphs2 = graphs
        for s, v in zip(self.states + self.enc_alt, saved):
            s.copy_(v)
        for o, v in zip(outs, kept):
            o[b:b + self._group + 2].copy_(v)
        self.t2.copy_(t0)
        self.te2.copy_(t1)

    def _run_overlap(self, frames):
        """Software pipeline over frames: E(0) | {D(0) || E(1)} | {H(0),E(2) || D(1)} | ... | H(last).  The trailing head is
        flushed before returning, so after run(n) all n frames are complete."""
        if frames <= 0:
            return
        if self.use_graph and self._graphs2 is None:
            self._capture_overlap()
        if self.use_graph and frames >= self.WHOLE_RUN_MIN:
            # a run length that comes again from the same frame parity (an event length; bench.py's --steps): graphs of its own
            key = ("seen", self._frames_done == 0, self._frames_done % 2, frames)
            self._run_seen[key] = self._run_seen.get(key, 0) + 1
            if self._run_seen[key] >= 2 and _tuning_env("URNN_TUNE_WHOLE_RUN", "1") != "0":
                return self._run_overlap_chunks(frames)
        i = 0
        while i < frames:
            t = self._frames_done
            if t == 0:                  # pipeline prologue: E(0)
                if self.use_graph:
                    self._graphs2["prologue"].replay()
                else:
                    self._enc_chain(0)
            first = i == 0              # no head pending at the start of a run() call
            if self.use_graph and not first and self._group and i + self._group <= frames:
                self._graphs2[("group", t % 2)].replay()
                self._frames_done += self._group
                i += self._group
                continue
            if self.use_graph:
                self._graphs2[("first", t % 2) if first else t % 2].replay()
            else:
                self._iter_overlap(t % 2, with_head=not first)
            self._frames_done += 1
            i += 1
        if self.use_graph:
            self._graphs2[("tail", (self._frames_done - 1) % 2)].replay()
        else:
            self._head_chain((self._frames_done - 1) % 2)

    CHUNK_FRAMES, CHUNK_GRAPHS = 120, 6

    def _run_overlap_chunks(self, frames):
        """run(frames) of the three-chain schedule as ONE graph replay per <= CHUNK_FRAMES frames instead of one per GROUP iterations (+ the
        first iteration, the single ones and the trailing head of their own): the same launches in the same order -- captured on demand,
        the first time a run length comes again (``_run_overlap``), the last few kept.  Every replay boundary is a join and a fork of the
        three streams and a graph launch: 2.1 % of the frames/s at 500x500, for a 20-frame call as for a 120-frame one
        (profiles/r06_whole_run_graphs.txt)."""
        t, end = self._frames_done, self._frames_done + frames
        nchunks = -(-frames // self.CHUNK_FRAMES)
        sizes = [frames // nchunks + (1 if k < frames % nchunks else 0) for k in range(nchunks)]
        for ci, sz in enumerate(sizes):
            first, last = ci == 0, ci == nchunks - 1
            key = ("chunk", t == 0, t % 2, first, last, sz)
            if key not in self._graphs2:
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    if t == 0:
                        self._enc_chain(0)                           # pipeline prologue E(0)
                    for k in range(sz):
                        self._iter_overlap((t + k) % 2, with_head=not (first and k == 0))
                    if last:
                        self._head_chain((t + sz - 1) % 2)           # the trailing head: after run(n) all n frames are complete
                chunks = [q for q in self._graphs2 if isinstance(q, tuple) and q[0] == "chunk"]
                if len(chunks) >= self.CHUNK_GRAPHS:
                    del self._graphs2[chunks[0]]                     # (dicts keep insertion order: the oldest)
                self._graphs2[key] = g
            self._graphs2[key].replay()
            t += sz
            self._frames_done = t
        assert t == end

    # -- level pipeline: four concurrent chains on small planes, a frame apart ------------------------------------
    # A small plane (64x64: 64 + 16 + 4 tiles of 64 pixels over the three resolutions) cannot fill the chip from one kernel chain, nor from
    # three: its launches are latency -- 18-30 us per cooperative cell, 13 us per 1x1 conv, whatever the size -- and a frame of the
    # three-chain schedule costs the LONGEST chain (the encoder pass: three cells + three convs, ~125-147 us at 64x64).  But only the
    # cells' own states recur from frame to frame: enc1(t+1) needs enc1(t) and nothing deeper, enc2(t+1) needs enc1(t+1) and enc2(t), ...
    # (encoder.py:119-215, decoder.py:102-217).  So the timestep is cut by LEVEL into units on FOUR streams (the runtime multiplexes
    # streams onto four hardware queues: seven streams measured 100 us per frame with two units sharing a queue, GPU_MAX_HW_QUEUES=8 three
    # times slower), and iteration i runs unit u on frame i - lag(u): a unit only ever reads what an EARLIER iteration wrote, and a frame
    # costs the longest unit (57-66 us at 64x64) + ~10 us per cross-queue wait instead of the longest chain.  What a unit hands to a later
    # one travels through rings of ``period`` buffers indexed by frame % period -- the encoder states (enc1(t) is read by dec1(t) three or
    # four iterations after it was written, when enc1 has moved on), dec3's and dec2's states (their deconvs may run an iteration later), the
    # stage outputs, the head's feature map; dec1's state stays in place.  The captured graphs exist per frame % period (even: the
    # frame-counter words alternate by frame parity): a pipeline fill (``depth`` iterations, units joining by lag), one steady iteration,
    # ``group`` steady iterations in one replay, and a drain.  run(n) leaves nothing in flight: n frames cost n + depth iterations.
    #
    # A plan: (stream, lag, launches) per unit; a launch that reads what an earlier one of the SAME frame wrote sits behind it in the same unit
    # or in a unit of larger lag.  "period" = buffers per ring; "group" = steady iterations per replay.
    #   F (forward, the default): four units in network order on streams 0..3, so every hand-over goes to a LATER stream -- inside a replay
    #     stream q + 1 waits for streams 0..q of the previous iteration and nothing waits for a later stream, the replays' ends are the only
    #     barriers.  Nothing but a replay's length bounds how far stream 0 runs ahead, so the rings hold group + depth + 1 frames.
    #   A / B: five units, a barrier between iterations (every stream waits for all four); B's balance (63 / 66 / 59 / 57 us at 64x64) was
    #     ahead at that one shape until a whole event became one replay; A is B with dec3 next to dec2 for planes whose enc1 is slow.  Kept
    #     as measured alternatives (URNN_TUNE_LEVEL_PLAN under URNN_TUNING=1) and in the tests.
    LEVEL_PLANS = level_schedule.PLANS          # (the schedule itself is data: level_schedule.py, model-checked by tests/test_level_schedule.py)
    LEVEL_STREAMS = level_schedule.STREAMS
    # batch x plane up to which the level pipeline is the default.  One event: ahead of three chains up to 448x448 (us per frame, three chains |
    # levels, both with their run-length graphs: 256x256 247 | 224, 288x288 288 | 298, 320x320 337 | 336, 384x384 477 | 417, 400x400 467 | 426,
    # 448x448 543 | 532; 400x560 585 | 615, 500x500 664 | 671 -- the big planes keep three chains).  Batched events: even at 128x128 x 8.
    LEVELS_MAX_PIXELS = 256 * 256
    LEVELS_MAX_PIXELS_ONE = 448 * 448

    def _lv_segments(self, u, tau):
        """The launches of unit ``u`` (of this engine's plan) for frame ``tau`` as closures (one ABI call each), buffers by tau % period."""
        P = self._lvP
        k, km, par = tau % P, (tau - 1) % P, tau % 2
        enc, dec = self.net.encoder, self.net.decoder
        E, R, D1, D2, ws = self._ring_e, self._ring, self._ring_d1, self._ring_d2, self._ws[u]
        d3 = self.states[5]

        def head():
            tail = self._stem_stats()
            self.net.head.run(R["feat"][k], out_masked=self.out_masked, out_cls=self.out_cls, out_raw=self.out_raw,
                              frame_index=self.t2[par:par + 1], ws=ws, partial0=self._k1part[k] if tail else None,
                              coop=self._head_coop and not tail, frame_next=self.t2[1 - par:2 - par])
        launch = {
            "stage1": lambda: self._stage1(self.te2[par:par + 1], t_next=self.te2[1 - par:2 - par]),
            "enc1": lambda: self._cell("enc1", enc.rnn1, self.a1, None, E[0][km], E[0][k], ws),
            "conv2": lambda: enc.stage2(E[0][k], out=R["a2"][k]),
            "enc2": lambda: self._cell("enc2", enc.rnn2, R["a2"][k], None, E[1][km], E[1][k], ws),
            "conv3": lambda: enc.stage3(E[1][k], out=R["a3"][k]),
            "enc3": lambda: self._cell("enc3", enc.rnn3, R["a3"][k], None, E[2][km], E[2][k], ws),
            "dec3": lambda: self._cell("dec3", dec.rnn3, None, E[2][k], D1[km], D1[k], ws),
            "deconv3": lambda: dec.stage3(D1[k], out=R["u3"][k]),
            "dec2": lambda: self._cell("dec2", dec.rnn2, R["u3"][k], E[1][k], D2[km], D2[k], ws),
            "deconv2": lambda: dec.stage2(D2[k], out=R["u2"][k]),
            "dec1": lambda: self._cell("dec1", dec.rnn1, R["u2"][k], E[0][k], d3, d3, ws),
            "lastconv": lambda: self._last_conv(d3, R["feat"][k], self._k1part[k]),
            "head": head,
        }
        return [launch[name] for name in self._plan[u][2]]

    def _run_iterations(self, its):
        """Iterations (i, lo, hi) of the level pipeline over the frames lo <= t < hi, back to back in one replay: the stream operations
        of level_schedule.events -- fork from the current stream, waits between iterations (a barrier, or forward only), the units'
        launches round-robin, join -- issued on this engine's streams (eagerly, or under capture)."""
        cur = torch.cuda.current_stream(self.device)
        stream = lambda q: cur if q == level_schedule.CUR else self._side[q]
        order = [int(c) for c in _tuning_env("URNN_TUNE_LEVEL_ORDER", "")] or None
        segs = {}
        for ev in level_schedule.events(self._plan_dict, its, order):
            if ev[0] == "wait":
                stream(ev[1]).wait_stream(stream(ev[2]))
            else:
                _, q, u, name, tau = ev
                if (u, tau) not in segs:
                    segs[(u, tau)] = dict(zip(self._plan[u][2], self._lv_segments(u, tau)))
                with torch.cuda.stream(self._side[q]):
                    segs[(u, tau)][name]()

    def _capture_levels(self):
        """Graphs of the level pipeline, per frame % period: ("fill", p), ("steady", p), ("group", p) = ``group`` steady iterations,
        ("drain", p).  As in _capture_overlap the warm-up and the first replay of every graph run for real -- on whatever the buffers hold,
        into output rows b .. b + 8 -- and states, rings, counters and those rows are put back afterwards."""
        NL = level_schedule.depth(self._plan_dict)
        self.net.head.flat_params()
        keep = [self.states[5]] + [t for ring in self._ring_e + [self._ring_d1, self._ring_d2] for t in ring] + [self.t2, self.te2]
        keep += [ws[:4] for ws in self._ws]         # (the status words: what the warm-up computes on stale buffers is not this event's)
        saved = [t.clone() for t in keep]
        rows = self.out_masked.shape[0]
        b = max(0, min(self._frames_done + (self._frames_done & 1), (rows - 8) & ~1))
        outs = [o for o in (self.out_masked, self.out_cls, self.out_raw) if o is not None]
        kept = [o[b:b + 8].clone() for o in outs]
        pair = torch.tensor([b, b + 1], dtype=torch.int32)

        def set_counters():
            self.t2.copy_(pair)
            self.te2.copy_(pair)
        set_counters()
        self._run_iterations([(i, 0, 2) for i in range(2 + NL)])     # warm-up (packs weights), eager: two frames through the whole pipeline
        torch.cuda.synchronize(self.device)
        plans = level_schedule.graph_plans(self._plan_dict)      # (frame numbers only matter modulo the period)
        graphs = {}
        for key, its in plans.items():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._run_iterations(its)
            graphs[key] = g
        for key in plans:                           # first launch = upload; one graph at a time (see _capture_overlap)
            set_counters()
            graphs[key].replay()
            torch.cuda.synchronize(self.device)
        self._graphs2 = graphs
        for t, v in zip(keep, saved):
            t.copy_(v)
        for o, v in zip(outs, kept):
            o[b:b + 8].copy_(v)
        torch.cuda.synchronize(self.device)

    WHOLE_RUN_MIN, WHOLE_RUN_MAX, WHOLE_RUN_GRAPHS = 12, 1024, 4

    def _run_levels(self, frames):
        if frames <= 0:
            return
        f = self._frames_done
        # A run length seen before from the same frame phase (an event length: every event of a data set has the reference's `duration`,
        # test.py:352) gets a graph of its own, captured the second time it comes and kept for the last few lengths: one replay with a
        # barrier every ``group`` iterations instead of fill + groups + single iterations + drain, ~18 us per replay end at 64x64.
        key = ("run", f % self._lvP, frames)
        whole = False
        if self.use_graph and self.WHOLE_RUN_MIN <= frames <= self.WHOLE_RUN_MAX:
            self._run_seen[key] = self._run_seen.get(key, 0) + 1
            whole = self._run_seen[key] >= 2
        for k, its in level_schedule.replays(self._plan_dict, f, frames, graphs=self.use_graph, whole=whole):
            if k is None:                           # eager, or a run shorter than the pipeline is deep (its fill and drain overlap)
                self._run_iterations(its)
                continue
            if self._graphs2 is None:               # fill | groups of steady iterations | single steady iterations | drain
                self._capture_levels()
            if k not in self._graphs2:              # (a whole run: captured on demand; its first replay uploads it)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._run_iterations(its)
                runs = [q for q in self._graphs2 if q[0] == "run"]
                if len(runs) >= self.WHOLE_RUN_GRAPHS:
                    del self._graphs2[runs[0]]      # (dicts keep insertion order: the oldest)
                self._graphs2[k] = g
            self._graphs2[k].replay()
        self._frames_done += frames

    def final_states(self):
        """The six states after the frames run so far (overlap mode keeps the newest encoder states in the buffer
        of the last frame's parity)."""
        if self.levels:
            k = (self._frames_done - 1) % self._lvP                  # (before the first frame: a slot reset() has zeroed)
            return [ring[k] for ring in self._ring_e] + [self._ring_d1[k], self._ring_d2[k], self.states[5]]
        if not self.overlap or self._frames_done == 0:
            return list(self.states)
        enc = self._enc_bufs((self._frames_done - 1) % 2)[1]
        return list(enc) + list(self.states[3:])

    def _capture(self):
        # warm-up on a side stream (also builds every packed-weight cache), then capture one timestep
        self.net.head.flat_params()
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(s):
            self._step()
        torch.cuda.current_stream(self.device).wait_stream(s)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self._step()
        self._graph = g
        # ... and ONE_CHAIN_GROUP timesteps in one replay (the frame counter is a device word the step itself advances): a graph launch per
        # frame leaves the chip idle for a few microseconds every frame
        self._graph_group = None
        self._one_group = max(1, int(_tuning_env("URNN_TUNE_ONE_CHAIN_GROUP", self.ONE_CHAIN_GROUP)))
        if self._one_group > 1:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(self._one_group):
                    self._step()
            self._graph_group = g

    ONE_CHAIN_GROUP = 8

    # -- public API -----------------------------------------------------------------------------------
    def load_event(self, event):
        """Copy an event (reference dict layout, or already flattened by event_to_device) into the static
        buffers.  Returns the number of frames T."""
        ev = event if "rain" in event else event_to_device(event, self.device)
        if ev["B"] != self.B or ev["H"] != self.H or ev["W"] != self.W:
            raise RuntimeError("event shape does not match the engine")
        T = ev["T"]
        if T > self.Tcap:
            raise RuntimeError(f"event has {T} frames, engine was built for {self.Tcap}")
        if (ev["rain"].dim() == 4) != self.spatial:
            raise RuntimeError("engine was built for %s rainfall" % ("spatial" if self.spatial else "scalar"))
        self.rain.zero_()
        self.cumsum.zero_()
        self.rain[:, :T].copy_(ev["rain"])
        self.cumsum[:, :T].copy_(ev["cumsum"])
        self.dem.copy_(ev["dem"])
        self.imperv.copy_(ev["imperv"])
        self.manhole.copy_(ev["manhole"])
        if self.S1 is not None:
            conv = self.net.encoder.stage1.layer
            self._w1.copy_(conv.weight.detach().reshape(conv.out_channels, -1))
            ops.stage1_static(self.dem, self.imperv, self.manhole, ev["dem_min"], ev["dem_max"], self._w1, self.nums, out=self.S1)
        self._check_params()
        if (ev["dem_min"], ev["dem_max"]) != (self.dem_min, self.dem_max):
            self.dem_min, self.dem_max = ev["dem_min"], ev["dem_max"]
            if self.S1 is None:
                # spatial rain: the bounds are arguments of the captured preprocess kernel -> re-capture when they change.
                # Scalar rain normalises the DEM in stage1_static above (eager, once per event): the graph does not see them.
                self._graph = None
                self._graphs2 = None
        return T

    def _check_params(self, walk=True):
        """A captured timestep holds raw pointers to the PACKED copies of the weights (and to the head's stacked affines), so a
        weight change must drop the graphs: ``load_state_dict`` / in-place edits bump the tensors' version counters, an
        optimizer that re-homes or rewrites parameters behind torch's back (``Trainer``) bumps ``net._urnn_generation``."""
        # ... and the launches inside a graph keep the GEMM arithmetic (urnn_set_matrix_mode) they were captured with
        # (the walk over the module tree is two thirds of this check's 60 us: reset() -- once per event -- walks, a run() in mid-event asks the
        # Parameter objects found then; replacing a Parameter OBJECT between two run() calls of one event goes through net._urnn_generation)
        gen = getattr(self.net, "_urnn_generation", 0)
        if walk or self._params is None or self._param_stamp is None or gen != self._param_stamp[0]:
            self._params = list(self.net.parameters())
        stamp = (gen, lib().urnn_get_matrix_mode()) + tuple((p.data_ptr(), p._version) for p in self._params)
        if stamp != self._param_stamp:
            self._tails = None                  # (a layer's `wide` flag may have changed with its weights)
            self._stem = None
            if self._param_stamp is not None:
                self._graph = None
                self._graphs2 = None
            self._param_stamp = stamp

    def reset(self):
        """Zero states and frame counter: the start of an event (test.py:352-356)."""
        self._check_params()
        self._zeros.zero_()
        self._counters.zero_()
        self._frames_done = 0

    def run(self, frames):
        """Roll ``frames`` timesteps from the current states / frame counter.  Asynchronous."""
        self._check_params(walk=False)        # a Trainer may have updated / re-homed the weights since the graphs were captured (mid-event run)
        if self.levels:
            return self._run_levels(frames)
        if self.overlap:
            return self._run_overlap(frames)
        self._frames_done += frames
        if self.use_graph:
            if self._graph is None:
                saved = [s.clone() for s in self.states]
                t0 = self.t_dev.clone()
                self._capture()
                for s, v in zip(self.states, saved):
                    s.copy_(v)
                self.t_dev.copy_(t0)
            left = frames
            while self._graph_group is not None and left >= self._one_group:
                self._graph_group.replay()
                left -= self._one_group
            for _ in range(left):
                self._graph.replay()
        else:
            for _ in range(frames):
                self._step()

    def rollout(self, event):
        """Full event from zero states: returns the (T,B,H,W) masked-depth frames (a view of the engine's
        output buffer, on device).  Raises FloatingPointError when a norm saw non-finite statistics (check_status)."""
        T = self.load_event(event)
        self.reset()
        self.run(T)
        self.check_status()
        return self.out_masked[:T]

    # -- operand-range guard (include/urnn_hip.h, "Operand range of the default matrix mode") ---------------------------
    def check_status(self):
        """Once per event: the kernels that fold norm statistics OR a bit into word 0 of their workspace when the sums are not
        finite -- an activation beyond the f16 pieces' range (|x| >= 2047) leaves the matrix pipe as inf and shows here.  Reading
        the word synchronises (the caller is about to fetch the frames anyway).  On a hit: re-run the first frames eagerly with a
        finiteness check behind every layer and raise naming the first one that fails."""
        bits = 0
        for word in self._ws_all.view(torch.int32)[::self._ws_stride // 4].cpu().tolist():
            bits |= int(word)
        if bits == 0:
            return
        for ws in self._ws:
            ws[:4].zero_()
        if bits & ops.STATUS_BARRIER:
            # not an arithmetic problem: a cooperative launch ran without all of its blocks resident and went on with incomplete
            # statistics.  The frames of this event are invalid; an engine built with coop_cells=False has no grid barriers.
            raise RuntimeError("U-RNN rollout: " + ops.STATUS_NAMES[ops.STATUS_BARRIER] + ".  The event's frames are invalid; "
                               "rebuild the engine with RolloutEngine(..., coop_cells=False) on this device.")
        what = "; ".join(name for bit, name in ops.STATUS_NAMES.items() if bits & bit)
        layer = self._first_nonfinite_layer()
        raise FloatingPointError(
            f"U-RNN rollout: non-finite {what}" + (f"; first layer with a non-finite output: {layer}" if layer else "") +
            ".  The default matrix mode carries operands as f16 pieces (|activation| < 2047, |weight| < 64, include/urnn_hip.h); "
            "run this checkpoint / event under ops.matrix_mode('fp32_mfma').")

    def _first_nonfinite_layer(self, max_frames=4):
        """Eager replay of the first frames from zero states with torch.isfinite behind every layer (diagnosis only)."""
        net = self.net
        enc, dec = net.encoder, net.decoder
        saved_t = int(self.t_dev.item())
        states = [torch.zeros_like(s) for s in self.states]
        e1, e2, e3, d1, d2, d3 = states
        t_dev = torch.zeros_like(self.t_dev)
        ws = ops.workspace(self._ws[0].numel(), self.device)
        found = None

        def bad(name, t):
            nonlocal found
            if found is None and not bool(torch.isfinite(t).all()):
                found = name
            return found is not None
        for _ in range(min(max_frames, self.Tcap)):
            self._stage1(t_dev)
            if bad("encoder.stage1", self.a1):
                break
            steps = (("encoder.rnn1", lambda: enc.rnn1.step(self.a1, None, e1, out=e1, ws=ws), e1),
                     ("encoder.stage2", lambda: enc.stage2(e1, out=self.a2), self.a2),
                     ("encoder.rnn2", lambda: enc.rnn2.step(self.a2, None, e2, out=e2, ws=ws), e2),
                     ("encoder.stage3", lambda: enc.stage3(e2, out=self.a3), self.a3),
                     ("encoder.rnn3", lambda: enc.rnn3.step(self.a3, None, e3, out=e3, ws=ws), e3),
                     ("decoder.rnn3", lambda: dec.rnn3.step(None, e3, d1, out=d1, ws=ws), d1),
                     ("decoder.stage3", lambda: dec.stage3(d1, out=self.u3), self.u3),
                     ("decoder.rnn2", lambda: dec.rnn2.step(self.u3, e2, d2, out=d2, ws=ws), d2),
                     ("decoder.stage2", lambda: dec.stage2(d2, out=self.u2), self.u2),
                     ("decoder.rnn1", lambda: dec.rnn1.step(self.u2, e1, d3, out=d3, ws=ws), d3),
                     ("decoder.stage1", lambda: dec.stage1(d3, out=self.feat), self.feat))
            for name, fn, out in steps:
                fn()
                if bad(name, out):
                    break
            if found:
                break
            masked, cls, _ = net.head.run(self.feat, ws=ws)
            if bad("head", masked) or bad("head", cls):
                break
            ops.advance_counter(t_dev, 1)
        self.t_dev.fill_(saved_t)
        return found
