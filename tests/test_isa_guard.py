"""Build-time guard of the band kernel's COUNTED wait (csrc/dense_band_body.h).

The threshold + corner band kernels synchronise a tile row with `s_waitcnt vmcnt(2 * BAND_DEPTH - 1)` + `s_barrier`
instead of draining the vector-memory queue: that is correct only while every wave issues, on EVERY path through one
loop iteration, its LDS-DMA of the tile row BAND_DEPTH ahead FIRST and then at least one store -- so that the DMA that is
waited for always has at least 2 * BAND_DEPTH - 1 younger operations.  hipcc has broken such an assumption once (it merged
three padding stores of the prologue into one: a race at 3840x2160), and nothing in the C++ says what
the compiler may do to the loop.  So this test compiles the kernel to ISA (device code only, no GPU needed) and checks,
for every instantiation, in the instruction stream itself:
  * each of the three unrolled iterations starts with exactly `s_waitcnt vmcnt(2 * BAND_DEPTH - 1) lgkmcnt(0)`, `s_barrier`;
  * the first vector-memory operation after the barrier is the LDS-DMA (`buffer_load_dwordx4 ... lds`), optionally
    followed by ONE conditional second DMA (the ninth chunk of a wide band) whose branch rejoins immediately, and then
    the flush store -- all before any other branch, i.e. unconditionally;
  * no `s_waitcnt vmcnt(0)` was inserted into the loop, except behind an atomic that returns a value or a spill reload
    (the candidate list's slot reservation: rare path) -- a drain elsewhere would not break correctness but would
    serialise the pipeline;
  * the prologue waits outright (`vmcnt(0)`) before the first barrier, and the kernel drains before it ends.
It fails if someone -- or a new hipcc -- changes the number or the order of vector-memory operations per iteration."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "robot_camera_calibration_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def listing():
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = os.path.join(CSRC, "k_dense_band.isa.s")
    deps = [os.path.join(CSRC, f) for f in ("k_dense_band.hip", "dense_band_body.h", "dense_rows.h", "rcc_internal.h")]
    if not os.path.exists(out) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(out):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                               "-o", out, os.path.join(CSRC, "k_dense_band.hip")], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


def _kernels(lines, stem="k_dense_band"):
    """{mangled name: [instruction / label lines]} for the kernels whose name starts with `stem`"""
    out, name = {}, None
    for l in lines:
        m = re.match(r"^(_Z\d+%s\S*):" % stem, l)
        if m:
            name = m.group(1); out[name] = []; continue
        if name is None:
            continue
        t = l.strip()
        if t.startswith(".Lfunc_end"):
            name = None; continue
        if not t or t.startswith(";") or (t.startswith(".") and not re.match(r"^\.LBB\d+_\d+:", t)):
            continue
        out[name].append(t)
    return out


def _is_dma(t): return t.startswith("buffer_load_dwordx4") and t.rstrip().endswith("lds")
def _is_store(t): return t.startswith(("buffer_store", "global_store", "flat_store"))
def _is_vmem(t): return t.startswith(("buffer_", "global_", "flat_", "scratch_"))
def _is_branch(t): return t.startswith(("s_cbranch", "s_branch"))
def _is_label(t): return re.match(r"^\.LBB\d+_\d+:", t) is not None


def _head_events(ins, b):
    """vector-memory operations every wave issues after the barrier at ins[b], in order, up to and including the first
    UNCONDITIONAL store: 'DMA' / 'ST' / 'VMEM:<op>' outside any branch, 'cDMA' for an LDS-DMA inside a short forward skip
    (the ninth chunk of a wide band).  A forward branch whose target label follows within 60 instructions opens a
    conditional region that ends at that label (scalar address selection compiles to such diamonds); any other branch
    ends the unconditional head."""
    ev, skip_to, j = [], None, b + 1
    while j < len(ins):
        t = ins[j]
        if skip_to is not None and t.startswith(skip_to + ":"):
            skip_to = None
        elif _is_branch(t):
            target = t.split()[-1]
            ahead = [k for k in range(j + 1, min(len(ins), j + 60)) if ins[k].startswith(target + ":")]
            if skip_to is None and ahead:
                skip_to = target
            elif skip_to is None:
                break                                   # a far / backward branch: the head is over
        elif _is_vmem(t):
            if skip_to is None:
                ev.append("DMA" if _is_dma(t) else "ST" if _is_store(t) else "VMEM:" + t.split()[0])
                if ev[-1] == "ST":
                    break
            elif _is_dma(t):
                ev.append("cDMA")
            else:
                ev.append("cVMEM:" + t.split()[0])
        elif t.startswith(("s_barrier", "s_endpgm")):
            break
        j += 1
    return ev


def test_counted_wait_invariant_holds_in_the_compiled_band_kernels(listing):
    depth = int(re.search(r"#define\s+BAND_DEPTH\s+(\d+)", open(os.path.join(CSRC, "dense_band_body.h")).read()).group(1))
    want_wait = "s_waitcnt vmcnt(%d) lgkmcnt(0)" % (2 * depth - 1)
    kernels = _kernels(listing)
    # every instantiation the product's launcher can pick: stage / compact forms x one-band / wide-band rings (the sweep-only
    # instantiations of the two-kernel form exist in the experiments build only)
    assert len(kernels) >= 4, sorted(kernels)
    for name, ins in kernels.items():
        bars = [i for i, t in enumerate(ins) if t.startswith("s_barrier")]
        assert len(bars) == 3, "%s: %d barriers (the loop is unrolled by three)" % (name, len(bars))
        for b in bars:
            assert ins[b - 1].replace("  ", " ") == want_wait, "%s: barrier preceded by '%s', not '%s'" % (name, ins[b - 1], want_wait)
            # ---- the unconditional head of the iteration: DMA [, conditional second DMA that rejoins at once], store
            ev = _head_events(ins, b)
            assert ev[:1] == ["DMA"] and "ST" in ev and all(e in ("DMA", "ST", "cDMA") for e in ev), "%s: iteration head is %s" % (name, ev)
        # ---- no drain inside the loop
        first, last = bars[0], bars[-1]
        for i, t in enumerate(ins):
            if not (t.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", t)):
                continue
            if i < first:
                continue                                                     # prologue: waits outright, by design
            # the candidate path (rare: a wave that found corners) reserves list slots with a returning atomic, and in the
            # register-capped compact kernel reloads a few spilled values there: both wait for their own result
            behind_atomic = any(("atomic" in u) or u.startswith("scratch_load") for u in ins[max(0, i - 24):i])
            before_end = any(u.startswith("s_endpgm") for u in ins[i:i + 8])
            assert behind_atomic or before_end, "%s: '%s' inside the loop (instruction %d; barriers at %s)" % (name, t, i, bars)
        # ---- prologue and epilogue drains exist
        assert any(t.startswith("s_waitcnt") and "vmcnt(0)" in t for t in ins[:first]), "%s: the prologue no longer waits outright" % name
        ends = [i for i, t in enumerate(ins) if t.startswith("s_endpgm")]
        assert ends and any(any(u.startswith("s_waitcnt") and "vmcnt(0)" in u for u in ins[max(0, e - 8):e]) for e in ends), \
            "%s: no drain of the in-flight DMA before the kernel ends" % name


def test_guard_catches_a_broken_iteration(listing):
    """the checker itself: drop the flush store of one iteration from a copy of the listing and it must object"""
    kernels = _kernels(listing)
    name, ins = sorted(kernels.items())[0]
    b = [i for i, t in enumerate(ins) if t.startswith("s_barrier")][1]
    j = next(i for i in range(b + 1, len(ins)) if _is_store(ins[i]))
    broken = ins[:j] + ins[j + 1:]
    ev = _head_events(broken, b)
    assert not (ev[:1] == ["DMA"] and "ST" in ev and all(e in ("DMA", "ST", "cDMA") for e in ev)), ev
    # ... and a store that only some waves issue (moved behind a short forward branch) does not count either
    k = next(i for i in range(b + 1, len(ins)) if _is_dma(ins[i]))
    cond = ins[:j] + ["s_cbranch_scc1 .LBB999_1", ins[j], ".LBB999_1:"] + ins[j + 1:]
    ev = _head_events(cond, b)
    assert not (ev[:1] == ["DMA"] and "ST" in ev and all(e in ("DMA", "ST", "cDMA") for e in ev)) or ev.index("ST") > 1, ev


# ---- the wave-per-window kernel (csrc/k_dense_wave.hip): the same counted wait, no barrier -------------------------------
def _wave_listing(exp):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = os.path.join(CSRC, "k_dense_wave_exp.isa.s" if exp else "k_dense_wave.isa.s")
    deps = [os.path.join(CSRC, f) for f in ("k_dense_wave.hip", "dense_wave_body.h", "dense_band_body.h", "dense_rows.h", "rcc_internal.h")]
    if not os.path.exists(out) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(out):
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only"] +
                              (["-DRCC_EXPERIMENTS"] if exp else []) + ["-o", out, os.path.join(CSRC, "k_dense_wave.hip")], stderr=subprocess.DEVNULL)
    return open(out).read().split("\n")


@pytest.fixture(scope="module", params=[False, True], ids=["product", "experiments"])
def wave_listing(request):
    """the product build carries ONE k_dense_wave (every window on its own); the gang form is instantiated only with
    -DRCC_EXPERIMENTS (librcc_hip_exp.so), where the same invariants are checked on both"""
    return _wave_listing(request.param), request.param


def test_counted_wait_invariant_holds_in_the_compiled_wave_kernel(wave_listing):
    wave_listing, exp_build = wave_listing
    """k_dense_wave: every wave is on its own, so a staged tile row is guarded by `s_waitcnt vmcnt(2 * WAVE_DEPTH - 1)` alone.
    That needs, per unrolled iteration and on every path: the wait, then the LDS-DMA of the row WAVE_DEPTH ahead as the
    first vector-memory operation, and ONE byte store of the tile levels after the corner stages have rejoined (more
    operations -- the candidate path -- only make the wait stricter; fewer would let it pass early)."""
    src = open(os.path.join(CSRC, "dense_wave_body.h")).read()
    depth = int(re.search(r"#define\s+WAVE_DEPTH\s+(\d+)", src).group(1))
    want_wait = "s_waitcnt vmcnt(%d) lgkmcnt(0)" % (2 * depth - 1)
    kernels = _kernels(wave_listing, "k_dense_wave")
    assert len(kernels) >= 1
    assert len(kernels) == (2 if exp_build else 1), sorted(kernels)   # every window on its own (the product's form) [+ gangs of eight: experiments build only]
    for name, ins in kernels.items():
        gang = re.search(r"ILi\d+ELi8EE", name) is not None
        bars = [i for i, t in enumerate(ins) if t.startswith("s_barrier")]
        if not gang:
            assert not bars, "%s: a barrier in the barrier-free kernel" % name
        else:
            # the gang's meeting point: one (conditional) barrier per unrolled iteration, and the counted wait still comes after
            # it -- nothing vector-memory in between, so the wait's arithmetic is untouched
            assert len(bars) == 3, "%s: %d barriers" % (name, len(bars))
            for b in bars:
                nxt = next(i for i in range(b + 1, len(ins)) if ins[i].replace("  ", " ") == want_wait)
                assert not any(_is_dma(t) or _is_store(t) for t in ins[b:nxt]), "%s: vector-memory operation between the barrier and the wait" % name
        waits = [i for i, t in enumerate(ins) if t.replace("  ", " ") == want_wait]
        stores = [i for i, t in enumerate(ins) if t.startswith("buffer_store_byte")]
        dmas = [i for i, t in enumerate(ins) if _is_dma(t)]
        assert len(waits) == 3 and len(stores) == 3, "%s: %d counted waits, %d level stores (the loop is unrolled by three)" % (name, len(waits), len(stores))
        assert len(dmas) == 3 + depth, "%s: %d LDS-DMA instructions" % (name, len(dmas))
        for n, wi in enumerate(waits):
            # the first vector-memory operation after the wait is the DMA, before any branch
            ev = _head_events(ins, wi)        # (scalar row-address selection compiles to short forward diamonds: followed)
            assert ev[:1] == ["DMA"], "%s: iteration %d starts with %s" % (name, n, ev)
            # exactly one level store before the next counted wait (or the loop's end), issued after the last join: no
            # branch between the nearest label above it and the store, so no path skips it
            nxt = waits[n + 1] if n + 1 < len(waits) else len(ins)
            mine = [si for si in stores if wi < si < nxt]
            assert len(mine) == 1, "%s: %d level stores in iteration %d" % (name, len(mine), n)
            k = mine[0]
            lab = max(i for i in range(wi, k) if _is_label(ins[i]))
            assert not any(_is_branch(t) for t in ins[lab:k]), "%s: the level store of iteration %d sits behind a branch" % (name, n)
        # no drain inside the loop except behind the candidate path's returning atomic; outright wait in the prologue, drain at the end
        for i, t in enumerate(ins):
            if t.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", t) and waits[0] < i:
                behind_atomic = any("atomic" in u for u in ins[max(0, i - 24):i])
                before_end = any(u.startswith("s_endpgm") for u in ins[i:i + 8])
                assert behind_atomic or before_end, "%s: '%s' inside the loop (instruction %d)" % (name, t, i)
        assert any(t.startswith("s_waitcnt") and "vmcnt(0)" in t for t in ins[dmas[0]:waits[0]]), "%s: the prologue no longer waits outright" % name
        ends = [i for i, t in enumerate(ins) if t.startswith("s_endpgm")]
        assert ends and any(any(u.startswith("s_waitcnt") and "vmcnt(0)" in u for u in ins[max(0, e - 8):e]) for e in ends)


# ---- register budgets of the step's kernels (round 4) ---------------------------------------------------------------------------
# A kernel's wave slots per SIMD follow from its register count (512 / allocation granule of 8), and two of the step's kernels are
# bound by how many ready waves a SIMD holds.  Round 4 lost one wave per SIMD in k_subpix without anyone noticing (78 -> 94
# registers when the gate moved into the candidate loop: 0.175 -> 0.195 ms, found in the rocprof summary of the NEXT profile set).
# This compiles the kernels for gfx950 (no GPU needed) and holds each to the occupancy the measurements in DESIGN.md section 5 were
# taken at.
BUDGETS = [("k_subpix.hip", r"_Z8k_subpixILb0E", 80, 6, "board-scene sub-pixel stage: six waves per SIMD"),
           ("k_dense_wave.hip", r"_Z12k_dense_waveILi0ELi1E", 80, 6, "threshold + corner pass of the step: six waves per SIMD (amdgpu_waves_per_eu)"),
           ("k_ingest.hip", r"_Z15k_ingest_stagedILi3ELb0ELi16E", 64, 8, "ingest pass: four 512-thread workgroups per CU")]


@pytest.mark.parametrize("src,symbol,max_vgprs,min_occupancy,what", BUDGETS, ids=[b[0] for b in BUDGETS])
def test_register_budget_of_step_kernels(tmp_path, src, symbol, max_vgprs, min_occupancy, what):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not found")
    out = str(tmp_path / (src + ".s"))
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                           "-I", os.path.join(ROOT, "include"), "-o", out, os.path.join(CSRC, src)], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    start = [i for i, l in enumerate(lines) if re.match(r"^%s\S*:" % symbol, l)]
    assert start, "kernel %s not found in the listing of %s" % (symbol, src)
    nv = occ = scratch = None
    for l in lines[start[0]:]:
        m = re.match(r"^; NumVgprs: (\d+)", l)
        if m and nv is None: nv = int(m.group(1))
        m = re.match(r"^; ScratchSize: (\d+)", l)
        if m and scratch is None: scratch = int(m.group(1))
        m = re.match(r"^; Occupancy: (\d+)", l)
        if m:
            occ = int(m.group(1)); break
    assert nv is not None and occ is not None, "no register summary behind %s" % symbol
    assert nv <= max_vgprs and occ >= min_occupancy and scratch == 0, "%s: %d registers, occupancy %d, scratch %s (%s)" % (symbol, nv, occ, scratch, what)
