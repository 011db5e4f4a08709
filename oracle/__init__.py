"""oracle/ - TEST INFRASTRUCTURE ONLY (checker of the HIP path): only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import anything from here; the product (icon_amd/) never does."""
