#!/usr/bin/env python3
"""Round 6: is the half-rate result transfer some runs see ("The result pipeline": the
8-byte-record leg of configs[3] at threshold 0.01 at 78 or at 134 ms per step, box by box) the state of the PCIe link?

Reads current_link_speed / current_link_width (and the maxima) of every AMD display / accelerator function and of its upstream
bridge from sysfs -- an ordinary user may read them --, then times a plain 1 GiB device-to-host copy into a long-lived page-locked
block: idle, right after two seconds of dense compute, and WHILE compute runs on another stream; link state again at the end.
Run on the GPU box: `python3 probes/pcie_link.py`."""
import glob
import json
import os
import time

import torch


def links():
    out = []
    for d in sorted(glob.glob("/sys/bus/pci/devices/*")):
        try:
            vendor = open(os.path.join(d, "vendor")).read().strip()
            cls = open(os.path.join(d, "class")).read().strip()
        except OSError:
            continue
        if vendor != "0x1002" or not (cls.startswith("0x03") or cls.startswith("0x12")):
            continue

        def rd(p, name):
            try:
                return open(os.path.join(p, name)).read().strip()
            except OSError:
                return None
        up = os.path.dirname(os.path.realpath(d))
        out.append({"bdf": os.path.basename(d), "class": cls,
                    "speed": rd(d, "current_link_speed"), "width": rd(d, "current_link_width"),
                    "max_speed": rd(d, "max_link_speed"), "max_width": rd(d, "max_link_width"),
                    "bridge": os.path.basename(up), "bridge_speed": rd(up, "current_link_speed"), "bridge_width": rd(up, "current_link_width")})
    return out


def d2h_gbs(dev, host, stream, reps=5):
    best = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            host.copy_(dev, non_blocking=True)
        stream.synchronize()
        best.append(dev.numel() * dev.element_size() / (time.perf_counter() - t0) / 1e9)
    return [round(x, 1) for x in best]


def main():
    rec = {"bus_id_of_device_0": getattr(torch.cuda.get_device_properties(0), "pci_bus_id", None), "links_before": links()}
    n = 1 << 30
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    host = torch.empty(n, dtype=torch.uint8, pin_memory=True)
    copy_stream = torch.cuda.Stream()
    rec["d2h_idle_GBps"] = d2h_gbs(dev, host, copy_stream)
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        b = a @ a
    torch.cuda.synchronize()
    rec["d2h_right_after_compute_GBps"] = d2h_gbs(dev, host, copy_stream)
    # ... and while compute runs
    during = []
    for _ in range(5):
        for _ in range(40):
            b = a @ a
        t1 = time.perf_counter()
        with torch.cuda.stream(copy_stream):
            host.copy_(dev, non_blocking=True)
        copy_stream.synchronize()
        during.append(round(n / (time.perf_counter() - t1) / 1e9, 1))
        torch.cuda.synchronize()
    rec["d2h_during_compute_GBps"] = during
    time.sleep(3.0)
    rec["d2h_after_3s_idle_GBps"] = d2h_gbs(dev, host, copy_stream)
    rec["links_after"] = links()
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
