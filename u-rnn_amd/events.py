"""Event loader for UrbanFlood24 / LarNO-style ``.npy`` event folders -- the storage half of the reference's
``Dynamic2DFlood`` dataset (Dynamic2DFlood.py:50-262), SURVEY.md 8(f) N1.  Same constructor arguments, same item layout
``[input_vars, target_vars, event_dir]``, so ``urnn_amd.inference.Inference`` (and the reference's own test loop) can be fed
from it; the per-frame feature assembly stays on the GPU (``urnn_amd.dataset.preprocess_inputs`` / ``RolloutEngine``).

Directory layout (Dynamic2DFlood.py:91-96,143-177)::

    <data_root>/{train,test}/geodata/<location>/{absolute_DEM,impervious,manhole}.npy        (H, W)
    <data_root>/{train,test}/flood/<location>/<event>/{flood,rainfall}.npy                   (T,H,W) m / (T,) or (T,H,W) mm

Only numpy and torch tensors on the host; nothing here touches the HIP library.
"""
import os

import numpy as np
import torch

from .distributed import shard_events


def _location_key(name):
    """Locations sort numerically when their name holds digits (location2 < location16), else by name
    (Dynamic2DFlood.py:98-103)."""
    digits = "".join(ch for ch in name if ch.isdigit())
    return (0, int(digits), "") if digits else (1, 0, name)


class Dynamic2DFlood(torch.utils.data.Dataset):
    """One sample = one rainfall event x one catchment location; raw (un-normalised) tensors.

    Deviation from the reference: it ships ``train.txt`` / ``test.txt`` event lists next to its module; here, when
    ``event_list_file`` is None and no ``<split>.txt`` sits next to this file, the events are the sorted sub-directories of
    the first location."""

    def __init__(self, data_root, split, event_list_file=None, duration=360, location=""):
        super().__init__()
        self.data_root = data_root
        self.duration = int(duration)
        self.data_dir = os.path.join(data_root, "train" if "train" in split else "test")
        self.geo_root = os.path.join(self.data_dir, "geodata")
        self.flood_root = os.path.join(self.data_dir, "flood")
        found = sorted(os.listdir(self.flood_root), key=_location_key)
        if location:
            if location not in found:
                raise ValueError(f"Location '{location}' not found in {self.flood_root}. Available: {found}")
            self.locations = [location]
        else:
            self.locations = found
        self.locations_dir = [os.path.join(self.flood_root, loc) for loc in self.locations]
        self.event_names = self._load_event_names(split, event_list_file)
        self.num_samples = len(self.event_names) * len(self.locations)

    def _load_event_names(self, split, event_list_file=None):
        path = event_list_file
        if path is None:
            beside = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{split}.txt")
            path = beside if os.path.isfile(beside) else None
        if path is None:
            first = self.locations_dir[0]
            return sorted(d for d in os.listdir(first) if os.path.isdir(os.path.join(first, d)))
        with open(path, "r") as fh:
            return [ln.strip() for ln in fh if ln.strip()]

    def __len__(self):
        return self.num_samples

    # -- raw files ------------------------------------------------------------------------------------
    def _event_dir(self, index):
        if not 0 <= index < self.num_samples:
            raise IndexError(index)
        nloc = len(self.locations)
        return os.path.join(self.flood_root, self.locations[index % nloc], self.event_names[index // nloc]), self.locations[index % nloc]

    @staticmethod
    def _load_dir(path, out):
        for name in sorted(os.listdir(path)):
            full = os.path.join(path, name)
            if os.path.isdir(full) or name.endswith(".jpg"):
                continue
            out[os.path.splitext(name)[0]] = np.load(full, allow_pickle=True)

    def _load_event(self, index):
        event_dir, loc = self._event_dir(index)
        data = {}
        self._load_dir(event_dir, data)                               # flood, rainfall
        self._load_dir(os.path.join(self.geo_root, loc), data)        # absolute_DEM, impervious, manhole
        return data, event_dir

    # -- model-ready tensors (Dynamic2DFlood.py:179-252) --------------------------------------------------
    def _prepare_input(self, event_data, event_dir=None, duration=None):
        T = self.duration if duration is None else int(duration)
        dem = torch.from_numpy(np.asarray(event_data["absolute_DEM"])).float() * 1000      # metres -> mm
        rain = torch.from_numpy(np.asarray(event_data["rainfall"])).float()
        if rain.shape[0] < T:                                         # zero-pad along time; longer series are kept whole
            pad = torch.zeros((T - rain.shape[0],) + tuple(rain.shape[1:]), dtype=rain.dtype)
            rain = torch.cat([rain, pad], 0)
        cums = torch.cumsum(rain, dim=0)
        if rain.ndim == 1:                                            # scalar rain (T,) -> (T,1,1,1)
            rain, cums = rain.reshape(-1, 1, 1, 1), cums.reshape(-1, 1, 1, 1)
        else:                                                         # spatial rain (T,H,W) -> (T,1,H,W)
            rain, cums = rain.unsqueeze(1), cums.unsqueeze(1)
        plane = lambda k: torch.from_numpy(np.asarray(event_data[k])).float()[None, None]
        dem = dem[None, None]
        return {
            "absolute_DEM": dem, "max_DEM": dem.max(), "min_DEM": dem.min(),
            "impervious": plane("impervious"), "manhole": plane("manhole"),
            "rainfall": rain, "cumsum_rainfall": cums,
        }

    def _prepare_target(self, event_data, duration=None):
        T = self.duration if duration is None else int(duration)
        flood = torch.from_numpy(np.asarray(event_data["flood"])).float()[:T] * 1000       # metres -> mm
        if flood.ndim == 4 and flood.shape[1] == 1:
            flood = flood.squeeze(1)
        return flood

    def __getitem__(self, index):
        data, event_dir = self._load_event(index)
        return [self._prepare_input(data, event_dir), self._prepare_target(data), event_dir]

    # -- inference helpers ----------------------------------------------------------------------------------
    def batched(self, index):
        """Item ``index`` with the leading batch dimension a ``DataLoader(batch_size=1)`` would add (the layout
        ``preprocess_inputs`` / ``Inference`` take)."""
        inp, tgt, event_dir = self[index]
        return {k: v.unsqueeze(0) for k, v in inp.items()}, tgt.unsqueeze(0), event_dir

    def shard(self, rank, world_size):
        """Sample indices of ``rank`` under ``DistributedSampler(shuffle=False)`` (test.py:741-746)."""
        return shard_events(len(self), rank, world_size)
