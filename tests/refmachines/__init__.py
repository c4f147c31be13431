"""Reference state machines kept for the tests only: the torch statement of
scipy's Nelder-Mead and the Python generator statement of scipy's BFGS, both
pinned to scipy by tests/test_tools_cpu.py; the product runs csrc/nm.hip and
csrc/bfgs_host.cpp, which the tests compare with these."""
