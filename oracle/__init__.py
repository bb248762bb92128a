"""CPU oracle for the model/dim3 forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cbim_amd`` imports this package; it
is imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` — as the checker / the timed CPU baseline,
never as the thing that is shipped or measured as the product.

The oracle is a plain restatement of the reference's algorithm with stock
``torch.nn.functional`` ops executed on the host CPU (fp32 or fp64).  Every
function cites the reference ``file:line`` it follows.  It is pinned against the
real reference by ``tests/golden/make_golden.py`` (which imports
``/root/reference`` in the build container and writes the fixtures the
``-m "not gpu"`` tests replay) and, when ``/root/reference`` is present, by a
direct comparison in ``tests/test_oracle.py``.
"""
