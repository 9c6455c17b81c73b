#!/usr/bin/env python3
"""Developer probe: the ordering tests of multi-device handles repeated in ONE process (two caller streams, two host
threads, partitioned vectors).  BSM_DIST_WORKERS / BSM_DIST_COPIES select the issue and the transfer path."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bsm_amd as bsm
from oracle import load_oracle
import test_gpu_multidevice as T
orc = load_oracle()
for it in range(12):
    T.test_unequal_parts_driven_from_two_streams(torch, bsm, orc)
    T.test_concurrent_products_on_one_multi_device_handle(torch, bsm, orc)
    if not os.environ.get("BSM_DIST_COPIES"):  # (the partitioned-vector entry has no copy path)
        T.test_partitioned_vectors_through_the_multi_device_handle(torch, bsm, orc, "symmetric", 4)
    print("soak round", it, "ok", flush=True)
