"""CPU oracle for the k-nearest-path scan -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package; shadowing_amd/ never does.  See psh_oracle.c for the
reference lines each function restates and for the parity-pinning status.
"""
from .oracle import (  # noqa: F401
    build, lib, qnorm, scan_topk, all_distances, gather_paths, shadow,
    scan_topk_embedded, all_distances_embedded, all_acc, all_acc_embedded,
)
