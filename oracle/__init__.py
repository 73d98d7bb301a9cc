"""CPU oracle for the WKV6 hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product package
(``rwkv_lm_ext_amd``) never does.
"""
