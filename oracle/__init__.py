"""oracle/ -- CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import, link or execute anything in this directory, and
only as the checker / the reported CPU baseline.  The product package
rvspecfit_amd/ never imports it.

Parity status: pinned against golden vectors captured from the reference
(tests/golden/) and, for the spline, against the reference's own C source
compiled in place (oracle/_ref/, build container only).
"""
