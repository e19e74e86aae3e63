"""CPU oracle for the tactics2d hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (tactics2d_amd) never does.
"""
