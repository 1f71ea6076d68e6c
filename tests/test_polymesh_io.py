"""polyMesh directory I/O (C++ host library, include/smhost.h): round trips in both formats, format
details OpenFOAM readers rely on, error reporting."""
import os

import numpy as np
import pytest


def _same(a, b, exact_points=True):
    for f in ("faceOffsets", "facePoints", "owner", "neighbour"):
        assert np.array_equal(getattr(a, f), getattr(b, f)), f
    assert a.nCells == b.nCells
    assert [(p.name, p.type, p.nFaces, p.startFace, p.myProcNo, p.neighbProcNo) for p in a.patches] == \
           [(p.name, p.type, p.nFaces, p.startFace, p.myProcNo, p.neighbProcNo) for p in b.patches]
    if exact_points:
        assert np.array_equal(a.points, b.points)


@pytest.mark.parametrize("binary", [False, True])
def test_roundtrip(tmp_path, binary):
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import read_polymesh, write_polymesh
    m = hex_block(4, 3, 2, jitter=0.2, seed=1)
    d = str(tmp_path / "constant" / "polyMesh")
    write_polymesh(d, m, binary=binary, precision=17)
    r = read_polymesh(d)
    _same(m, r)
    txt = open(os.path.join(d, "faces"), "rb").read()
    assert (b"faceCompactList" in txt) == binary and (b"faceList" in txt) != binary
    assert b"nInternalFaces:" in open(os.path.join(d, "owner"), "rb").read()


def test_processor_patches_and_addressing(tmp_path):
    from smoothmesh_amd.meshgen import hex_subdomain
    from smoothmesh_amd.polymesh import read_label_list, read_polymesh, write_decomposed_case
    subs = [hex_subdomain((3, 2, 2), (2, 1, 1), r, jitter=0.1) for r in range(2)]
    write_decomposed_case(str(tmp_path), subs)
    for s in subs:
        d = str(tmp_path / f"processor{s.rank}" / "constant" / "polyMesh")
        _same(s.mesh, read_polymesh(d))
        assert np.array_equal(read_label_list(os.path.join(d, "pointProcAddressing")), s.pointProcAddressing)
    b = open(tmp_path / "processor0" / "constant" / "polyMesh" / "boundary").read()
    assert "procBoundary0to1" in b and "neighbProcNo    1;" in b


def test_points_precision_and_time_instance(tmp_path):
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import read_polymesh, write_points, write_polymesh
    m = hex_block(3, jitter=0.3, seed=2)
    d = str(tmp_path / "constant" / "polyMesh")
    write_polymesh(d, m)
    moved = m.points + 1e-3
    write_points(str(tmp_path / "5" / "polyMesh"), moved, "5/polyMesh", precision=10)     # SM.C:2425
    r = read_polymesh(d, pointsDir=str(tmp_path / "5" / "polyMesh"))
    assert np.max(np.abs(r.points - moved)) / np.max(np.abs(moved)) < 1e-9   # 10 significant digits
    assert 'location    "5/polyMesh"' in open(tmp_path / "5" / "polyMesh" / "points").read()


@pytest.mark.parametrize("binary", [False, True])
def test_gzip_compressed_files(tmp_path, binary):
    """writeCompression on: every file is <file>.gz; a reader takes <file>.gz where <file> is missing -- checked
    against files compressed by an independent writer (Python's gzip) and against our own writer"""
    import gzip
    from smoothmesh_amd.meshgen import hex_block
    from smoothmesh_amd.polymesh import read_polymesh, set_write_compression, write_points, write_polymesh
    m = hex_block(4, 3, 3, jitter=0.2, seed=3)
    d = tmp_path / "constant" / "polyMesh"
    write_polymesh(str(d), m, binary=binary, precision=17)
    for name in ("points", "faces", "owner", "neighbour", "boundary"):
        raw = (d / name).read_bytes()
        with gzip.open(d / (name + ".gz"), "wb") as g:
            g.write(raw)
        os.remove(d / name)
    _same(m, read_polymesh(str(d)))
    # our writer: compressed output only, decodable by gzip, and no stale plain copy next to it
    set_write_compression(True)
    try:
        moved = m.points * 2.0
        t = tmp_path / "3" / "polyMesh"
        write_points(str(t), moved, "3/polyMesh", binary=binary, precision=17)
    finally:
        set_write_compression(False)
    assert sorted(os.listdir(t)) == ["points.gz"]
    assert b"vectorField" in gzip.open(t / "points.gz").read()
    assert np.array_equal(read_polymesh(str(d), pointsDir=str(t)).points, moved)
    # switching compression off again replaces the .gz
    write_points(str(t), moved, "3/polyMesh", binary=binary, precision=17)
    assert sorted(os.listdir(t)) == ["points"]


def test_reads_handwritten_openfoam_style(tmp_path):
    """comments, inline short lists, uniform-list shorthand, label=64 header -- as OpenFOAM writes them"""
    from smoothmesh_amd.polymesh import read_polymesh
    d = tmp_path / "constant" / "polyMesh"
    os.makedirs(d)
    hdr = 'FoamFile\n{\n version 2.0;\n format ascii;\n arch "LSB;label=32;scalar=64";\n class %s;\n location "constant/polyMesh";\n object %s;\n}\n// * * * //\n'
    (d / "points").write_text(hdr % ("vectorField", "points") + "8\n(\n(0 0 0) (1 0 0) (1 1 0) (0 1 0)\n(0 0 1) (1 0 1) /* c */ (1 1 1) (0 1 1)\n)\n")
    (d / "faces").write_text(hdr % ("faceList", "faces") + "6\n(\n4(0 4 7 3)\n4(1 2 6 5)\n4(0 1 5 4)\n4(3 7 6 2)\n4(0 3 2 1)\n4(4 5 6 7)\n)\n")
    (d / "owner").write_text(hdr % ("labelList", "owner") + "6{0}\n")
    (d / "neighbour").write_text(hdr % ("labelList", "neighbour") + "0()\n")
    (d / "boundary").write_text(hdr % ("polyBoundaryMesh", "boundary") +
                                "1\n(\n walls\n {\n type wall;\n inGroups 1(wall);\n nFaces 6;\n startFace 0;\n }\n)\n")
    m = read_polymesh(str(d))
    assert m.nPoints == 8 and m.nFaces == 6 and m.nInternalFaces == 0 and m.nCells == 1
    assert m.patches[0].type == "wall" and m.patches[0].nFaces == 6
    assert np.array_equal(m.owner, np.zeros(6, np.int32))


def test_errors_name_the_file(tmp_path):
    from smoothmesh_amd.polymesh import read_polymesh
    with pytest.raises(RuntimeError, match="points"):
        read_polymesh(str(tmp_path))


def test_obj_readers_agree_and_follow_openfoam_conventions(tmp_path):
    """constant/geometry/*.obj (SM.C:1924-1926): polygons become triangle fans about their first vertex, `l` records
    become consecutive pairs, unused points of an edge mesh are dropped, v/vt/vn references and negative indices work;
    the front-end's C++ reader and the Python reader return the same arrays"""
    from smoothmesh_amd.polymesh import read_obj
    from smoothmesh_amd.surfgen import read_obj_edges, read_obj_surface
    f = tmp_path / "s.obj"
    f.write_text("# comment\nmtllib x.mtl\no Cube\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0.5 0.5 1\nvn 0 0 1\nusemtl m\ns 0\n"
                 "f 1//1 2//1 3//1 4//1\nf 1/1/1 2/2/1 5/3/1\nf -1 -2 -3\n")
    p, t = read_obj_surface(str(f))
    assert p.shape == (5, 3) and t.tolist() == [[0, 1, 2], [0, 2, 3], [0, 1, 4], [4, 3, 2]]
    p2, t2 = read_obj(str(f), "surface")
    assert np.array_equal(p, p2) and np.array_equal(t, t2)
    g = tmp_path / "e.obj"
    g.write_text("v 9 9 9\nv 0 0 0\nv 1 0 0\nv 2 0 0\nv 7 7 7\nv 2 1 0\nl 2 3 4\nl 4 6\n")
    p, e = read_obj_edges(str(g))
    assert p.tolist() == [[0, 0, 0], [1, 0, 0], [2, 0, 0], [2, 1, 0]] and e.tolist() == [[0, 1], [1, 2], [2, 3]]
    p2, e2 = read_obj(str(g), "edges")
    assert np.array_equal(p, p2) and np.array_equal(e, e2)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(Exception, match="out of range"):
        read_obj(str(bad), "surface")


@pytest.mark.skipif(not os.path.isdir("/root/reference/testcase3/constant/geometry"), reason="reference tree not present")
def test_obj_readers_on_the_reference_test_cases():
    """the geometry files of the reference's own test cases (Blender / VTK exports with mtllib, o, vn, s records and quads)"""
    import glob
    from smoothmesh_amd.polymesh import read_obj
    from smoothmesh_amd.surfgen import read_obj_edges, read_obj_surface
    files = sorted(glob.glob("/root/reference/testcase*/constant/geometry/*.obj"))
    assert len(files) >= 10
    for f in files:
        kind = "surface" if f.endswith("targetSurfaces.obj") else "edges"
        a = (read_obj_surface if kind == "surface" else read_obj_edges)(f)
        b = read_obj(f, kind)
        assert len(a[1]) > 0 and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), f
        assert a[1].min() == 0 and a[1].max() == len(a[0]) - 1 if kind == "edges" else a[1].max() < len(a[0])


# ---- the several-thread readers / writers (round 6): the same bytes and the same numbers as the serial loops ------------------
def _run_py(code, env):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r)\n" % root + code], capture_output=True, text=True,
                       env=dict(os.environ, **env), timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def _nasty_doubles(n, seed=7):
    """every kind of value the %.{p}g formatter and the decimal parser treat differently: random bit patterns over the whole
    exponent range, subnormals, exact integers, halves (round-half-even ties at low precision), powers of ten and their
    neighbours, +-0, huge and tiny magnitudes"""
    rng = np.random.default_rng(seed)
    bits = rng.integers(0, 2**63, size=n, dtype=np.uint64) | (rng.integers(0, 2, size=n, dtype=np.uint64) << np.uint64(63))
    v = bits.view(np.float64).copy()
    v = v[np.isfinite(v)]
    extra = [0.0, -0.0, 1.0, -1.0, 0.5, 0.25, 0.125, 1.5, 2.5, 1e-5, 1e-4, 9.9999e-5, 0.0001, 123456789.0, 1234567890123456789.0, 1e15, 1e16,
             1e17, 1e21, 1e22, 1e23, 9007199254740992.0, 9007199254740993.0, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
             0.1, 0.2, 0.3, 1 / 3, 2 / 3, 1e-10, 0.30000000000000004, 99999.5, 999999.5, 9999995.0, 0.00012345678901234567]
    extra += [10.0 ** k for k in range(-30, 31)] + [np.nextafter(10.0 ** k, 0) for k in range(-20, 21)] + [np.nextafter(10.0 ** k, np.inf) for k in range(-20, 21)]
    near = rng.uniform(-2, 2, size=n)             # coordinates of a real mesh
    return np.concatenate([v, np.array(extra), -np.array(extra), near, np.round(near, 3), np.round(near * 1000)])


@pytest.mark.parametrize("precision", [1, 6, 10, 12, 17])
def test_threaded_point_writer_is_printf_byte_for_byte(tmp_path, precision):
    """writePoints formats with std::to_chars(general, precision) on several threads; the reference's mesh.write() is printf's
    %.{precision}g per coordinate (SM.C:2425 sets the precision): the file must be what a one-thread printf loop writes, byte for
    byte, for every kind of value"""
    from smoothmesh_amd.polymesh import write_points
    v = _nasty_doubles(20000)
    v = v[: (len(v) // 3) * 3].reshape(-1, 3)
    np.save(tmp_path / "v.npy", v)
    code = ("import numpy as np; from smoothmesh_amd.polymesh import write_points; "
            f"write_points({str(tmp_path / 'T')!r}, np.load({str(tmp_path / 'v.npy')!r}), 'T', precision={precision})")
    _run_py(code, {"SMHOST_IO_THREADS": "7", "SMHOST_IO_GRAIN": "500"})
    got = open(tmp_path / "T" / "points", "rb").read()
    body = got[got.index(b"\n(\n") + 3: got.rindex(b"\n)\n") + 1].decode()
    want = "".join("(%s %s %s)\n" % tuple("%.*g" % (precision, x) for x in row) for row in v)     # Python's % is C's printf
    assert len(body) == len(want) and body == want
    # ... and with one thread (the same code without the cutting)
    code1 = code.replace("'T'", "'S'").replace(str(tmp_path / "T"), str(tmp_path / "S"))
    _run_py(code1, {"SMHOST_IO_THREADS": "1"})
    assert open(tmp_path / "S" / "points", "rb").read().replace(b'"S"', b'"T"') == got


def test_threaded_ascii_readers_equal_the_serial_ones(tmp_path):
    """a whole polyMesh directory read with the lists cut into many pieces == read by the serial scanner == what was written;
    17 digits round-trip every coordinate exactly through the decimal parser (fast path and strtod path alike)"""
    from smoothmesh_amd.polymesh import cavity_mesh, write_polymesh
    m = cavity_mesh(8, jitter=0.2, seed=5)
    m.points[:, :] = _nasty_doubles(m.points.size, seed=11)[: m.points.size].reshape(-1, 3)      # (readers do not look at geometry)
    d = str(tmp_path / "constant" / "polyMesh")
    write_polymesh(d, m, binary=False, precision=17)
    code = ("import numpy as np; from smoothmesh_amd.polymesh import read_polymesh; "
            f"r = read_polymesh({d!r}); np.savez({str(tmp_path)!r} + '/out_' + TAG, p=r.points, fo=r.faceOffsets, fp=r.facePoints, o=r.owner, n=r.neighbour)")
    _run_py("TAG = 'par'; " + code, {"SMHOST_IO_THREADS": "13", "SMHOST_IO_GRAIN": "300"})
    _run_py("TAG = 'ser'; " + code, {"SMHOST_IO_THREADS": "1"})
    a, b = np.load(tmp_path / "out_par.npz"), np.load(tmp_path / "out_ser.npz")
    for k in ("p", "fo", "fp", "o", "n"):
        assert np.array_equal(a[k], b[k]), k
    assert np.array_equal(a["p"].view(np.uint64), m.points.view(np.uint64))         # bit for bit, -0.0 included
    assert np.array_equal(a["fo"], m.faceOffsets) and np.array_equal(a["fp"], m.facePoints)
    assert np.array_equal(a["o"], m.owner) and np.array_equal(a["n"], m.neighbour)


def test_decimal_parser_is_strtod(tmp_path):
    """the point reader's decimal -> double conversion (exact fast path for short mantissas, strtod for the rest) gives Python's
    float() -- a correctly rounded conversion like glibc's strtod -- for every spelling OpenFOAM or a user may write"""
    from smoothmesh_amd.polymesh import read_polymesh, write_polymesh
    from smoothmesh_amd.meshgen import hex_block
    m = hex_block(2, jitter=0.0)
    d = str(tmp_path / "constant" / "polyMesh")
    write_polymesh(d, m)
    toks = ["0", "-0", "+1", "1.", ".5", "-.5e1", "1e0", "1E+2", "1e-2", "123456789012345678", "1234567890123456789012", "0.1", "0.10000000000000001",
            "9007199254740993", "9007199254740992e1", "1e22", "1e23", "1e-22", "1e-23", "4.9e-324", "2.4703282292062328e-324", "1.7976931348623157e308",
            "0.000000000000000000000000000001", "100000000000000000000000", "3.14159265358979323846264338327950288", "1e400", "-1e400", "1e-400",
            "nan", "inf", "-inf", "12345.678e-3", "00012.5000", "5e-1", "7.0e+00"]
    rng = np.random.default_rng(3)
    for _ in range(3000):
        mant = "".join(rng.choice(list("0123456789"), size=rng.integers(1, 22)))
        cutp = rng.integers(0, len(mant) + 1)
        t = ("-" if rng.random() < 0.3 else "") + (mant[:cutp] or "0") + ("." + mant[cutp:] if cutp < len(mant) else "")
        if rng.random() < 0.5:
            t += "e%+d" % rng.integers(-330, 330)
        toks.append(t)
    # the extended-precision path (17 .. 19 digit mantissas, |exponent| <= 27): coordinates as precision 17 writes them, 19-digit
    # mantissas, and values constructed to sit next to a midpoint between two doubles (where only strtod may decide)
    toks += ["%.17g" % x for x in rng.uniform(-2, 2, 120000)] + ["%.16e" % x for x in rng.uniform(-1e3, 1e3, 40000)]
    for _ in range(40000):
        mant = "".join(rng.choice(list("0123456789"), size=19)).lstrip("0") or "1"
        toks.append(("-" if rng.random() < 0.5 else "") + mant[0] + "." + mant[1:] + "e%+d" % rng.integers(-9, 9))
    for x in rng.uniform(0.5, 2, 4000):
        lo = np.float64(x)
        hi = np.nextafter(lo, np.inf)
        from fractions import Fraction
        mid = (Fraction(float(lo)) + Fraction(float(hi))) / 2                  # exactly between two doubles
        for k in (-3, -1, 0, 1, 3):                                           # ... and a few units of the 19th digit around it
            v = mid + Fraction(k, 10 ** 18)
            digits = str(int(v * 10 ** 18))
            toks.append(digits[:-18] + "." + digits[-18:])
    while len(toks) % 3:
        toks.append("0")
    n = len(toks) // 3
    body = "".join("(%s %s %s)\n" % tuple(toks[3 * i: 3 * i + 3]) for i in range(n))
    head = open(os.path.join(d, "points")).read()
    head = head[: head.index("\n27\n")]
    with open(os.path.join(d, "points"), "w") as f:
        f.write(head + "\n%d\n(\n" % n + body + ")\n\n// ***** //\n")
    from smoothmesh_amd.polymesh import lib, _check, _p
    import ctypes as C
    want = np.array([float(t) for t in toks])
    for env in ({"SMHOST_IO_THREADS": "5", "SMHOST_IO_GRAIN": "200"}, {"SMHOST_IO_THREADS": "1"}):
        code = ("import numpy as np, ctypes as C; from smoothmesh_amd import polymesh as pm; L = pm.lib(); "
                f"L.smhost_read_points.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]; n = C.c_int64(-1); "
                f"assert L.smhost_read_points({os.path.join(d, 'points')!r}.encode(), None, C.byref(n)) == 0, L.smhost_last_error(); "
                f"a = np.empty(n.value); assert L.smhost_read_points({os.path.join(d, 'points')!r}.encode(), a.ctypes.data_as(C.POINTER(C.c_double)), C.byref(n)) == 0; "
                f"np.save({str(tmp_path / 'got.npy')!r}, a)")
        _run_py(code, env)
        got = np.load(tmp_path / "got.npy")
        assert got.shape == want.shape
        same = (got.view(np.uint64) == want.view(np.uint64)) | (np.isnan(got) & np.isnan(want))
        assert same.all(), [(toks[i], got[i], want[i]) for i in np.nonzero(~same)[0][:5]]


def test_odd_and_broken_ascii_lists_threaded_equals_serial(tmp_path):
    """OpenFOAM's ascii lists may carry comments, any white space, uniform forms; files get truncated.  Whatever the several-thread
    readers cannot take they must hand to the serial scanner: for every mutant of a small case both settings either return the
    same arrays or raise the same message -- and neither crashes the process or allocates for a count the file cannot hold"""
    import json
    from smoothmesh_amd.polymesh import cavity_mesh, write_polymesh
    m = cavity_mesh(6, jitter=0.2, seed=3)
    base = tmp_path / "base" / "constant" / "polyMesh"
    write_polymesh(str(base), m, binary=False, precision=12)
    rng = np.random.default_rng(2024)
    files = {n: open(base / n, "rb").read() for n in ("points", "faces", "owner", "neighbour", "boundary")}

    def body_span(b):
        i = b.index(b"\n(\n") + 3
        return i, b.rindex(b"\n)\n")

    mutants = []            # (name, file, bytes)
    for name in ("points", "faces", "owner", "neighbour"):
        b = files[name]
        i, j = body_span(b)
        mid = i + (j - i) // 2
        nl = b.index(b"\n", mid) + 1
        mutants += [
            (f"{name}: line comment inside", name, b[:nl] + b"// a remark (with brackets)\n" + b[nl:]),
            (f"{name}: block comment inside", name, b[:nl] + b"/* a remark\n over lines ) */\n" + b[nl:]),
            (f"{name}: tabs and CRLF", name, b[:i] + b[i:j].replace(b"\n", b"\r\n").replace(b" ", b"\t ") + b[j:]),
            (f"{name}: one line", name, b[:i] + b[i:j].replace(b"\n", b" ") + b[j:]),
            (f"{name}: truncated", name, b[:mid]),
            (f"{name}: truncated at a record", name, b[:nl]),
            (f"{name}: garbage byte", name, b[:nl] + b"x" + b[nl:]),
            (f"{name}: stray closer", name, b[:nl] + b")\n" + b[nl:]),
            (f"{name}: stray opener", name, b[:nl] + b"(\n" + b[nl:]),
            (f"{name}: no footer", name, b[:j + 3]),
            (f"{name}: nothing after the closer", name, b[:j + 2]),
            (f"{name}: count one too many", name, None),
            (f"{name}: count huge", name, None),
            (f"{name}: count negative", name, None),
            (f"{name}: one record short", name, b[:b.rindex(b"\n", i, j) + 1] + b[j + 1:]),
        ]
        for _ in range(6):           # random single-byte damage inside the body
            k = int(rng.integers(i, j))
            c = bytes([int(rng.choice(list(b" ()\n-.e9x/;{")))])
            mutants.append((f"{name}: byte {k - i} -> {c!r}", name, b[:k] + c + b[k + 1:]))
    # the counts: the line before "(\n"
    fixed = []
    for label, name, data in mutants:
        if data is None:
            b = files[name]
            i, _ = body_span(b)
            k = b.rindex(b"\n", 0, i - 3) + 1
            n = int(b[k:i - 3])
            new = {"count one too many": n + 1, "count huge": 10 ** 15, "count negative": -3}[label.split(": ")[1]]
            data = b[:k] + str(new).encode() + b[i - 3:]
        fixed.append((label, name, data))
    p = files["points"]
    i, j = body_span(p)
    first = p.index(b"\n", i) + 1
    fixed += [
        ("points: four numbers in a record", "points", p[:i] + b"(0 0 0 0)\n" + p[first:]),
        ("points: two numbers in a record", "points", p[:i] + b"(0 0)\n" + p[first:]),
        ("points: numbers glued by signs", "points", p[:i] + b"(1-2+3)\n" + p[first:]),
        ("points: nan and inf", "points", p[:i] + b"(nan inf -inf)\n" + p[first:]),
        ("points: hex float", "points", p[:i] + b"(0x1p-3 1e400 1e-400)\n" + p[first:]),
        ("owner: uniform list", "owner", files["owner"][:files["owner"].rindex(b"\n", 0, body_span(files["owner"])[0] - 3) + 1] + b"%d{0}\n" % (len(m.owner),)),
    ]
    cases = []
    for k, (label, name, data) in enumerate(fixed):
        d = tmp_path / f"m{k}" / "constant" / "polyMesh"
        os.makedirs(d)
        for n, b in files.items():
            open(d / n, "wb").write(data if n == name else b)
        cases.append((label, str(d)))
    json.dump(cases, open(tmp_path / "cases.json", "w"))
    code = f"""
import json, hashlib, numpy as np
from smoothmesh_amd.polymesh import read_polymesh
out = []
for label, d in json.load(open({str(tmp_path / 'cases.json')!r})):
    try:
        r = read_polymesh(d)
        h = hashlib.sha256()
        for a in (r.points, r.faceOffsets, r.facePoints, r.owner, r.neighbour):
            h.update(np.ascontiguousarray(a).tobytes())
        out.append([label, 'ok', h.hexdigest(), int(r.nPoints), int(r.nFaces)])
    except Exception as e:
        out.append([label, 'error', str(e)])
json.dump(out, open({str(tmp_path)!r} + '/out_' + TAG + '.json', 'w'))
"""
    _run_py("TAG = 'par'\n" + code, {"SMHOST_IO_THREADS": "11", "SMHOST_IO_GRAIN": "64"})
    _run_py("TAG = 'ser'\n" + code, {"SMHOST_IO_THREADS": "1"})
    a = json.load(open(tmp_path / "out_par.json"))
    b = json.load(open(tmp_path / "out_ser.json"))
    assert len(a) == len(b) == len(cases)
    for x, y in zip(a, b):
        assert x == y, (x, y)
    verdict = {x[0]: x[1] for x in a}
    # what must be read, what must be refused
    for name in ("points", "faces", "owner", "neighbour"):
        for ok in ("line comment inside", "block comment inside", "tabs and CRLF", "one line", "no footer", "nothing after the closer"):
            assert verdict[f"{name}: {ok}"] == "ok", (name, ok, [x for x in a if x[0] == f"{name}: {ok}"])
        for bad in ("truncated", "truncated at a record", "garbage byte", "stray closer", "count one too many", "count huge", "count negative", "one record short"):
            assert verdict[f"{name}: {bad}"] == "error", (name, bad)
    good = [x for x in a if x[0] == "points: line comment inside"][0]
    for name in ("points", "faces", "owner", "neighbour"):
        for ok in ("line comment inside", "block comment inside", "tabs and CRLF", "one line", "no footer"):
            assert [x for x in a if x[0] == f"{name}: {ok}"][0][2:] == good[2:]          # the same mesh as the undamaged files
    assert verdict["points: four numbers in a record"] == "error" and verdict["points: two numbers in a record"] == "error"
    assert verdict["points: nan and inf"] == "ok" and verdict["points: hex float"] == "ok" and verdict["owner: uniform list"] in ("ok", "error")
