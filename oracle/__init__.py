"""CPU oracle for the block-sparse mul! hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from .oracle import Oracle, load_oracle  # noqa: F401
