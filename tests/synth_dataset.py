"""Writes a tiny synthetic UrbanFlood24-style folder tree (seeded) -- shared by tests/golden/make_golden.py (which runs
the reference's Dynamic2DFlood over it) and tests/test_events_metrics.py (which runs ours over the same files)."""
import os

import numpy as np

LOCATIONS = ("location2", "location16", "region_b")     # numeric sort puts location2 before location16
EVENTS = ("rain_a", "rain_b")
H, W, T_FILE, DURATION = 8, 12, 5, 7                    # events shorter than the duration exercise the zero padding


def write_tree(root, seed=1234):
    rs = np.random.RandomState(seed)
    base = os.path.join(root, "test")
    for li, loc in enumerate(LOCATIONS):
        geo = os.path.join(base, "geodata", loc)
        os.makedirs(geo, exist_ok=True)
        np.save(os.path.join(geo, "absolute_DEM.npy"), rs.uniform(0, 10, (H, W)).astype(np.float32))
        np.save(os.path.join(geo, "impervious.npy"), rs.uniform(0, 1, (H, W)).astype(np.float32))
        np.save(os.path.join(geo, "manhole.npy"), (rs.uniform(0, 1, (H, W)) > 0.9).astype(np.float32))
        for ev in EVENTS:
            d = os.path.join(base, "flood", loc, ev)
            os.makedirs(d, exist_ok=True)
            spatial = li == 2                                   # the third location carries spatial rainfall
            rain = rs.uniform(0, 3, (T_FILE, H, W) if spatial else (T_FILE,)).astype(np.float32)
            np.save(os.path.join(d, "rainfall.npy"), rain)
            flood = rs.uniform(0, 0.4, (DURATION + 2, 1, H, W) if li == 1 else (DURATION + 2, H, W)).astype(np.float32)
            np.save(os.path.join(d, "flood.npy"), flood)
    lst = os.path.join(root, "events.txt")
    with open(lst, "w") as fh:
        fh.write("rain_b\n\nrain_a\n")
    return lst
