"""CPU oracle for the BN254 MSM/NTT hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  See bn254_oracle.c (C restatement) and pyref.py (big-integer twin).
"""
