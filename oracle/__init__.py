"""CPU oracle for the field_interpolation hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (field_interpolation_amd) never does.
"""
