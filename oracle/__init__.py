"""CPU oracle for the MIPSFusion render-and-optimise hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``mipsfusion_amd`` (the product) may import
this package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and only as the checker.

Contents
--------
tcnn_cpu.py     restatement of tiny-cuda-nn 1.7 HashGrid / Frequency / Identity
                encodings (third-party, absent from /root/reference:
                ``environment.yaml:74`` pins ``tinycudann==1.7``).
                *** parity unpinned *** -- the reference holds no test or golden
                vector at this boundary and tinycudann cannot be built here.
p3d_cpu.py      restatement of the pytorch3d quaternion helpers the reference
                imports (``helper_functions/geometry_helper.py:3-4``), also unpinned.
path_cpu.py     torch-CPU restatement of scene_rep / decoder / losses / Adam,
                pinned against the reference's own Python imported in the build
                container (``oracle/ref_import.py`` -> ``tests/golden/*.npz``).
ref_import.py   build-container-only shim that imports /root/reference's modules
                to generate the golden fixtures.  Never used on the GPU box.
c/              plain-C restatement of the hash-grid index/interpolation arithmetic
                (bit-level cross-check of tcnn_cpu.py, built by ``build()``).
"""
