"""TEST INFRASTRUCTURE ONLY.

CPU restatements of the FastEGNN hot path, used as the checker for the HIP
product path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; nothing under ``fastegnn_amd/``
does.
"""
