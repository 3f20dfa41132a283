"""oracle/ -- CPU restatements of the reference hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product (ray-marching_amd/, include/) never does.  "parity unpinned" against
the real wgpu render: see the header of rm_oracle.c.
"""
