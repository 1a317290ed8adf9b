"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the MSDeformAttn hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package, and only as the checker.  ``devis_amd`` never imports it.
"""
