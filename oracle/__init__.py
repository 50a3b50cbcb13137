"""ORACLE -- TEST INFRASTRUCTURE ONLY.

CPU restatements of the reference algorithms on the hot path, used as the checker by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.
Nothing under ``fv3net_amd/`` may import this package.
"""
