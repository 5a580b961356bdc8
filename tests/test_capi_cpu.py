"""-m "not gpu": the C-ABI library loads and exports every symbol include/ssqp_hip.h declares; host-side entry
points (generator, Phase-1) behave; the host mirror of the reference's types behaves.  No compute on a GPU."""
import ctypes as C
import hashlib
import os
import re
import warnings

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "ssqp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ssqp_[A-Za-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg._capi.lib()
    names = declared_functions()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert set(names) == set(pkg._capi.SIGNATURES), set(names) ^ set(pkg._capi.SIGNATURES)
    assert b"gfx950" in lib.ssqp_version()


def test_default_settings_match_reference(pkg):
    s = pkg._capi.CSettings()
    pkg._capi.lib().ssqp_default_settings(C.byref(s))
    assert (s.maxIter, s.tol, s.tolG) == (7777, 2.0 ** -26, 2.0 ** -33)      # types.jl:401-408


def test_no_cpu_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.NoDeviceError):
        pkg.Context(0)
    V = np.eye(3)
    with pytest.raises(pkg.NoDeviceError):
        pkg.solveQP(pkg.QP(V))


def test_generator_is_deterministic(pkg):
    cfg = pkg.GenConfig(16, 2, 3, 32, 1e-3, 0.2, 1.1, 0.1)
    a = pkg.generate_batch(cfg, 2, 42, nthreads=1)
    b = pkg.generate_batch(cfg, 2, 42, nthreads=2)
    for k in a:
        assert np.array_equal(a[k], b[k])
    assert np.array_equal(a["V"][0], a["V"][0].T)                            # exactly symmetric
    h = hashlib.sha256(b"".join(a[k].tobytes() for k in "VAGqbgdu")).hexdigest()
    assert h == GEN_HASH, h
    assert np.linalg.eigvalsh(a["V"][1])[0] > 0
    assert np.allclose(a["A"][0][:, 0], 1.0) and a["b"][0][0] == 1.0         # budget row


GEN_HASH = "735db9a7c0700096f60dc81b82fa349435c63430273719e8e08e8f7e3a183c1a"


def test_phase1_matches_oracle(pkg, orc):
    for cfg, seed in [(pkg.GenConfig(40, 1, 0, 80, 1e-3, 3 / 32, 1.2, 0.0), 1),
                      (pkg.GenConfig(64, 1, 6, 128, 1e-3, 0.07, 0.97, 0.1), 2),
                      (pkg.GenConfig(48, 3, 5, 96, 1e-3, 0.1, 1.0, 0.1), 3)]:
        prob = pkg.generate_batch(cfg, 6, 1000 + seed)
        x0, S0, st = pkg.phase1_batch(prob, nthreads=2)
        xo, So, sto = orc.initQP_batch(prob["A"], prob["G"], prob["b"], prob["g"], prob["d"], prob["u"])
        assert np.array_equal(st, sto) and np.array_equal(S0, So) and np.array_equal(x0, xo)
        assert (st == 1).all()
        # a vertex: #IN structurals == M + #active inequalities
        for p in range(6):
            assert (S0[p][:cfg.N] == 0).sum() == cfg.M + (S0[p][cfg.N:] == 4).sum()


def test_phase1_infeasible_and_unsupported_rule(pkg):
    cfg = pkg.GenConfig(20, 1, 0, 40, 1e-3, 0.01, 1.0, 0.0)
    prob = pkg.generate_batch(cfg, 1, 5)
    x0, S0, st = pkg.phase1_batch(prob)
    assert st[0] == 0
    with pytest.raises(pkg.SSQPError):
        pkg.phase1_batch(prob, settingsLP=pkg.Settings(rule="stpEdgeLP"))


def test_qp_constructor_mirrors_reference(pkg):
    V = np.array([[2.0, 1.0], [0.0, 2.0]])
    Q = pkg.QP(V)
    assert np.array_equal(Q.V, [[2, 0.5], [0.5, 2]]) and (Q.N, Q.M, Q.J, Q.mc) == (2, 1, 0, 1)   # types.jl:243
    assert np.array_equal(Q.A, np.ones((1, 2))) and Q.b.tolist() == [1.0] and Q.d.tolist() == [0, 0]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert pkg.QP(-np.eye(2)).mc == -70                                                      # types.jl:246-249
        assert pkg.QP(np.eye(2), d=np.array([0.0, 1.0]), u=np.array([1.0, 1.0])).mc == -30       # types.jl:275-278
        assert pkg.QP(np.eye(2), d=np.full(2, -np.inf)).mc == -20                                # types.jl:281-284
        d, u = np.array([0.0, 2.0]), np.array([1.0, 1.0])
        Q = pkg.QP(np.eye(2), d=d, u=u)                                                          # types.jl:286-292
        assert d.tolist() == [0.0, 1.0] and u.tolist() == [1.0, 2.0] and Q.mc == 1
    with pytest.raises(pkg.DimensionMismatch):
        pkg.QP(np.eye(3), A=np.ones((1, 2)))
    with pytest.raises(pkg.DimensionMismatch):
        pkg.QP(np.eye(3), q=np.zeros(2))
    with pytest.raises(TypeError):
        pkg.Settings(bogus=1)                                                                    # MOIwrapper.jl:17-31
    assert [int(s) for s in (pkg.IN, pkg.DN, pkg.UP, pkg.OE, pkg.EO)] == [0, 1, 2, 3, 4]        # types.jl:17-23


LLVM = "/opt/rocm/lib/llvm/bin"


def _device_disassemblies():
    """(object path, [(address, instruction text, branch target address or None)]) of every built device object of the library"""
    import glob
    import shutil
    import subprocess
    import tempfile
    objs = [o for o in sorted(glob.glob(os.path.join(ROOT, "statusswitchingqp.jl_amd", "csrc", "ssqp_*.o")))
            if "_prof" not in o and "_diag" not in o and not o.endswith("ssqp_host.o")]
    built = os.path.exists(os.path.join(ROOT, "statusswitchingqp.jl_amd", "libssqp_hip.so"))
    if not os.path.exists(os.path.join(LLVM, "llvm-objdump")):
        pytest.skip("no llvm tools in this image")
    # (a box that has the library but not its objects -- the GPU box gets both -- cannot run this; the build box must)
    assert objs or not built, "libssqp_hip.so is built but its device objects are gone: run `make` in csrc/"
    if not objs:
        pytest.skip("library not built")
    out = []
    tmp = tempfile.mkdtemp()
    try:
        for o in objs:
            fb, co = os.path.join(tmp, "fb.bin"), os.path.join(tmp, "dev.co")
            subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", o, fb], check=True)
            un = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                                 "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], capture_output=True)
            if un.returncode != 0:
                continue  # (an object without device code)
            dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout
            ins = []
            for ln in dis.splitlines():
                m = re.match(r"\s+(\S.*?)\s*// ([0-9A-F]{12}):", ln)
                if not m:
                    continue
                a, text, tgt = int(m.group(2), 16), m.group(1).strip(), None
                b = re.match(r"s_c?branch\w*\s+(\d+)$", text)
                if b:  # SOPP branch: simm16 dwords relative to the next instruction
                    off = int(b.group(1))
                    tgt = a + 4 + 4 * (off - 65536 if off >= 32768 else off)
                ins.append((a, text, tgt))
            out.append((o, ins))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def test_m0_belongs_to_the_lds_dma_loads():
    """The LDS-DMA loads of the big-factor build (glds16_s in ssqp_wave.hip) set M0 and do not restore it (LLVM ignores
    a clobber of the reserved m0): nothing else in the library's kernels may use M0.  Checked on the code objects of
    the built device objects."""
    n_dma = n_load = 0
    for o, ins3 in _device_disassemblies():
        ins = [s for _, s, _ in ins3]
        for i, s in enumerate(ins):
            if s.startswith("global_load_lds_dwordx4"):
                n_load += 1
            if not re.search(r"\bm0\b", s):
                continue
            assert re.fullmatch(r"s_mov_b32 m0, s\d+", s), (o, s)
            assert ins[i + 1].startswith("s_nop") and ins[i + 2].startswith("global_load_lds_dwordx4"), (o, ins[i:i + 3])
            n_dma += 1
    assert n_dma == n_load > 0


def test_no_store_inside_a_counted_wait_ring_loop():
    """The LDS ring of the big-factor build is consumed behind COUNTED waits (`s_waitcnt vmcnt(N)`, N > 0: "all but the
    N youngest vector-memory operations are done"), which is only a statement about the ring's LDS-DMA loads while
    nothing else shares the counter inside the loop: loads return in order, but a STORE -- a register spill the
    compiler placed there, say -- completes out of order with respect to them, and the wait could then pass with a ring
    slot still in flight (ssqp_wave.hip, "the column ring").  So: no store instruction inside any innermost loop of a
    wave build that holds an LDS-DMA load and a counted wait.  (The streamed delete stores on purpose and therefore
    waits with vmcnt(0).)"""
    store = re.compile(r"(scratch_store|global_store|buffer_store|flat_store|global_atomic|flat_atomic|buffer_atomic)")
    n_loops = 0
    for o, ins in _device_disassemblies():
        if "ssqp_wave" not in os.path.basename(o):
            continue
        index = {a: i for i, (a, _, _) in enumerate(ins)}
        loops = []
        for i, (a, s, tgt) in enumerate(ins):
            if re.match(r"s_c?branch", s) and tgt is not None and tgt <= a:
                assert tgt in index, (o, s, hex(tgt))
                loops.append((index[tgt], i))
        starts = sorted(loops)
        for b, e in loops:
            if any(b2 >= b and e2 <= e and (b2, e2) != (b, e) for b2, e2 in starts):
                continue  # not innermost
            body = [s for _, s, _ in ins[b:e + 1]]
            if not any(s.startswith("global_load_lds") for s in body):
                continue
            if not any(re.search(r"vmcnt\([1-9]\d*\)", s) for s in body):
                continue
            n_loops += 1
            bad = [s for s in body if store.match(s)]
            assert not bad, (o, hex(ins[b][0]), bad[:3])
    assert n_loops > 100, n_loops  # (268 in the round-3 objects)
