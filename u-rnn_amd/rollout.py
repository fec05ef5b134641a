"""On-device rollout engine: the T-step loop of ``test.Inference`` (test.py:326-377) with everything resident
in HBM -- six hidden states updated in place, per-frame input assembly, the whole timestep (about 45 kernel
launches) captured once as a hipGraph and replayed T times while a device-side frame counter advances.
No host round trip inside the loop; one D2H of the (T,B,H,W) result at the end if the caller wants numpy."""
import torch

from . import ops
from ._lib import lib
from .dataset import event_to_device
from .general import initialize_states


class RolloutEngine:
    def __init__(self, net, input_height, input_width, historical_nums, rain_max, cumsum_rain_max, batch=1,
                 max_frames=360, spatial_rain=False, net_cfg=None, use_graph=True, keep_raw=False,
                 device=None):
        self.net = net
        self.H, self.W = int(input_height), int(input_width)
        self.nums = int(historical_nums)
        self.C = 2 * self.nums + 3
        self.rain_max, self.cumsum_max = float(rain_max), float(cumsum_rain_max)
        self.B = int(batch)
        self.Tcap = int(max_frames)
        self.spatial = bool(spatial_rain)
        self.use_graph = bool(use_graph)
        self.device = torch.device(device) if device is not None else next(net.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("RolloutEngine needs the model on a GPU (HIP) device")
        self.net_cfg = net_cfg
        B, H, W, dev = self.B, self.H, self.W, self.device
        f32 = dict(dtype=torch.float32, device=dev)
        # static event buffers (filled per event by load_event)
        rshape = (B, self.Tcap, H, W) if self.spatial else (B, self.Tcap)
        self.rain = torch.zeros(rshape, **f32)
        self.cumsum = torch.zeros(rshape, **f32)
        self.dem = torch.zeros((B, H, W), **f32)
        self.imperv = torch.zeros((B, H, W), **f32)
        self.manhole = torch.zeros((B, H, W), **f32)
        self.dem_min, self.dem_max = 0.0, 1.0
        # recurrent state, updated in place
        self.states = list(initialize_states(dev, H, W, net_cfg, batch=B))
        # activations between kernels
        enc, dec = net.encoder, net.decoder
        self.x_in = torch.empty((B, self.C, H, W), **f32)
        self.a1 = torch.empty((B, enc.stage1.out_channels, H, W), **f32)
        self.a2 = torch.empty((B, enc.stage2.out_channels, H // 2, W // 2), **f32)
        self.a3 = torch.empty((B, enc.stage3.out_channels, H // 4, W // 4), **f32)
        self.u3 = torch.empty((B, dec.stage3.out_channels, H // 2, W // 2), **f32)
        self.u2 = torch.empty((B, dec.stage2.out_channels, H, W), **f32)
        self.feat = torch.empty((B, dec.stage1.out_channels, H, W), **f32)
        # outputs for every frame
        self.out_masked = torch.zeros((self.Tcap, B, H, W), **f32)
        self.out_cls = torch.zeros((self.Tcap, B, H, W), **f32)
        self.out_raw = torch.zeros((self.Tcap, B, H, W), **f32) if keep_raw else None
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.zero_frame = torch.zeros(1, dtype=torch.int32, device=dev)
        # scratch: size for the largest consumer, before any capture
        L = lib()
        need = max([L.urnn_head_workspace_bytes(B, 16, H, W)] +
                   [L.urnn_gru_cell_workspace_bytes(B, c.num_features, c.shape[0], c.shape[1])
                    for c in (enc.rnn1, enc.rnn2, enc.rnn3, dec.rnn3, dec.rnn2, dec.rnn1)])
        ops.WORKSPACE.reserve(need, dev)
        self._graph = None
        self._dem_stamp = None

    # -- one timestep, all launches on the current stream ----------------------------------------------
    def _step(self):
        net = self.net
        enc, dec = net.encoder, net.decoder
        e1, e2, e3, d1, d2, d3 = self.states
        ops.preprocess(self.rain, self.cumsum, self.dem, self.imperv, self.manhole, self.dem_min, self.dem_max, 0,
                       self.nums, self.rain_max, self.cumsum_max, out=self.x_in, t_dev=self.t_dev)
        enc.stage1(self.x_in, out=self.a1)
        enc.rnn1.step(self.a1, None, e1, out=e1)
        enc.stage2(e1, out=self.a2)
        enc.rnn2.step(self.a2, None, e2, out=e2)
        enc.stage3(e2, out=self.a3)
        enc.rnn3.step(self.a3, None, e3, out=e3)
        dec.rnn3.step(None, e3, d1, out=d1)
        dec.stage3(d1, out=self.u3)
        dec.rnn2.step(self.u3, e2, d2, out=d2)
        dec.stage2(d2, out=self.u2)
        dec.rnn1.step(self.u2, e1, d3, out=d3)
        dec.stage1(d3, out=self.feat)
        net.head.run(self.feat, out_masked=self.out_masked, out_cls=self.out_cls, out_raw=self.out_raw,
                     frame_index=self.t_dev)
        ops.advance_counter(self.t_dev, 1)

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
        if (ev["dem_min"], ev["dem_max"]) != (self.dem_min, self.dem_max):
            # normalisation bounds are kernel arguments frozen into the graph: re-capture when they change
            self.dem_min, self.dem_max = ev["dem_min"], ev["dem_max"]
            self._graph = None
        return T

    def reset(self):
        for s in self.states:
            s.zero_()
        self.t_dev.zero_()

    def run(self, frames):
        """Roll ``frames`` timesteps from the current states / frame counter.  Asynchronous."""
        if self.use_graph:
            if self._graph is None:
                saved = [s.clone() for s in self.states]
                t0 = self.t_dev.clone()
                self._capture()
                for s, v in zip(self.states, saved):
                    s.copy_(v)
                self.t_dev.copy_(t0)
            for _ in range(frames):
                self._graph.replay()
        else:
            for _ in range(frames):
                self._step()

    def rollout(self, event):
        """Full event from zero states: returns the (T,B,H,W) masked-depth frames (a view of the engine's
        output buffer, on device)."""
        T = self.load_event(event)
        self.reset()
        self.run(T)
        return self.out_masked[:T]
