"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the interior-point KKT path of omuses/hqp.  Nothing under
``oracle/`` may be imported by the product package ``hqp_amd``; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg use it.
"""
