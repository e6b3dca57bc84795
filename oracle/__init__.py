"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement of the reference's hot-path arithmetic.  Only tests/, bench.py's
``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import this package, and only as
the checker.  Nothing under jtsm_amd/ imports it (tests/test_layout.py enforces that).
"""
