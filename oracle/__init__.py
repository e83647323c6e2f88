"""CPU oracle for the StreamMOS streaming-inference path.

THIS PACKAGE IS TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker.  Nothing under
``streammos_amd/`` imports it; the product path fails loudly without its HIP library.

Parity pinning: every function here is a CPU restatement of a reference function (cited
file:line in its docstring) and is pinned by the golden vectors in ``tests/golden/*.npz``,
which were produced by running the real reference on CPU in the build container
(``tests/golden/make_golden.py``; the recipe is ``oracle/ref_import.py``).
"""
