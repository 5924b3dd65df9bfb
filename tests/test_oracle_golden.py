"""Pins oracle/abft_oracle.c against the committed golden fixtures
(tests/golden/, generated from the reference build by make_golden.py) and
against the code's own algebraic properties.  Runs anywhere (no GPU, no
reference tree)."""
import json
import os

import numpy as np
import pytest

from _oracle import COO, CSR, MODES, Oracle, OracleMatrix, event_lines

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FMT = {"csr": CSR, "coo": COO}
NW = {CSR: 3, COO: 4}
EW = {CSR: 2, COO: 0}


@pytest.fixture(scope="module")
def ecc():
    return json.load(open(os.path.join(G, "ecc.json")))


@pytest.fixture(scope="module")
def kern():
    return np.load(os.path.join(G, "kernels.npz"))


def words_of(fmt, vb, col, row):
    if fmt == CSR:
        return np.array([vb & 0xFFFFFFFF, vb >> 32, col], dtype=np.uint32)
    return np.array([col, row, vb & 0xFFFFFFFF, vb >> 32], dtype=np.uint32)


@pytest.mark.parametrize("name", ["csr", "coo"])
def test_generated_masks_equal_reference_constants(ecc, name):
    fmt = FMT[name]
    want = np.array([[int(v, 16) for v in r] for r in ecc[name]["masks"]], dtype=np.uint32)
    assert np.array_equal(Oracle.masks(fmt)[:, :NW[fmt]], want)


def test_survey_known_answers():
    """SURVEY 8c table (reference output)."""
    kats = [(0x4010000000000000, 0x000000, 0x00000000, 0xBE000000, 0xBE000000),
            (0xBFF0000000000000, 0x000001, 0x00000001, 0xAE000001, 0xAF000001),
            (0xBFF3C0CA428C59FB, 0xABCDEF, 0x00ABCDEF, 0xB2ABCDEF, 0xB2ABCDEF),
            (0x0000000000000000, 0xFFFFFF, 0x00FFFFFF, 0x00FFFFFF, 0x00FFFFFF),
            (0x01A56E1FC2F8F359, 0x003039, 0x80003039, 0xA4003039, 0xA4003039),
            (0x400921FB54442D18, 0x98967F, 0x8098967F, 0xD298967F, 0xD398967F)]
    L = Oracle.lib()
    for vb, col, sed, sec7, sec8 in kats:
        assert L.ora_csr_encode_col(2, vb, col) == sed
        assert L.ora_csr_encode_col(3, vb, col) == sec7
        assert L.ora_csr_encode_col(4, vb, col) == sec8
        assert L.ora_csr_encode_col(5, vb, col) == sec8
    vb, row, col = 0xBFF3C0CA428C59FB, 0x123456, 0xABCDEF
    assert L.ora_coo_encode_col(2, col, row, vb) == 0x80ABCDEF
    assert L.ora_coo_encode_col(3, col, row, vb) == 0xC0ABCDEF
    assert L.ora_coo_encode_col(4, col, row, vb) == 0xC1ABCDEF
    assert L.ora_coo_encode_col(5, col, row, vb) == 0xC1ABCDEF


@pytest.mark.parametrize("name", ["csr", "coo"])
def test_golden_encodes(ecc, name):
    fmt = FMT[name]
    for k in ecc[name]["kats"]:
        w = words_of(fmt, int(k["value_bits"], 16), k["col"], k["row"])
        for mode in MODES:
            assert int(Oracle.encode(fmt, mode, w)[EW[fmt]]) == k["encoded"][mode], (k, mode)


@pytest.mark.parametrize("name", ["csr", "coo"])
def test_golden_single_flip_table_and_decode(ecc, name):
    fmt = FMT[name]
    for s in ecc[name]["single_flips"]:
        e = np.zeros(NW[fmt], dtype=np.uint32)
        e[s["bit"] // 32] = 1 << (s["bit"] % 32)
        assert Oracle.syndrome(fmt, e) == s["syndrome"]
        assert Oracle.parity(fmt, e) == s["parity"] == 1
        if s["decoded"] is not None:
            assert Oracle.flipped_bit(fmt, s["syndrome"]) == s["decoded"] == s["bit"]
    for h, bit in ecc[name]["decode"].items():
        syn = sum(((int(h) >> (p - 1)) & 1) << (32 - p) for p in range(1, 8))
        assert Oracle.flipped_bit(fmt, syn) == bit


@pytest.mark.parametrize("fmt", [CSR, COO])
def test_exhaustive_single_and_double_flips(fmt):
    """every single flip trips parity and decodes to itself,
    except the overall-parity bit (syndrome 0); every double flip gives
    parity 0 and a non-zero syndrome."""
    nb = 32 * NW[fmt]
    pbit = 32 * EW[fmt] + 24
    rng = np.random.default_rng(3)
    for _ in range(3):
        w = rng.integers(0, 2**32, size=NW[fmt], dtype=np.uint64).astype(np.uint32)
        w[EW[fmt]] &= 0x00FFFFFF
        enc = Oracle.encode(fmt, "secded", w)
        assert Oracle.syndrome(fmt, enc) == 0 and Oracle.parity(fmt, enc) == 0
        for b in range(nb):
            f = enc.copy()
            f[b // 32] ^= np.uint32(1 << (b % 32))
            assert Oracle.parity(fmt, f) == 1
            s = Oracle.syndrome(fmt, f)
            if b == pbit:
                assert s == 0
            else:
                assert Oracle.flipped_bit(fmt, s) == b
            for b2 in range(b + 1, nb):
                g = f.copy()
                g[b2 // 32] ^= np.uint32(1 << (b2 % 32))
                assert Oracle.parity(fmt, g) == 0 and Oracle.syndrome(fmt, g) != 0


@pytest.mark.parametrize("mat", ["lap9x7", "rnd80", "lap16"])
@pytest.mark.parametrize("name", ["csr", "coo"])
@pytest.mark.parametrize("mode", MODES)
def test_golden_spmv_and_cg(kern, mat, name, mode):
    fmt = FMT[name]
    cols, rows, vals = kern[mat + "_cols"], kern[mat + "_rows"], kern[mat + "_vals"]
    n = int(kern[mat + "_n"][0])
    key = "%s_%s_%s" % (mat, name, mode)
    o = OracleMatrix(fmt, mode, cols, rows, vals, n)
    assert np.array_equal(o.stored_words(), kern[key + "_words"])
    y = o.spmv(kern[mat + "_x"])
    assert np.array_equal(y.view(np.uint64), kern[key + "_y"].view(np.uint64))
    it, hist, x, fatal = o.cg(kern[mat + "_b"])
    assert not fatal and it == len(kern[key + "_rr"])
    assert np.array_equal(hist.view(np.uint64), kern[key + "_rr"].view(np.uint64))
    assert np.array_equal(x.view(np.uint64), kern[key + "_xsol"].view(np.uint64))


def test_golden_flip_cases(kern):
    cases = json.load(open(os.path.join(G, "flips.json")))
    assert len(cases) > 200
    for c in cases:
        mat, fmt = c["matrix"], FMT[c["fmt"]]
        n = int(kern[mat + "_n"][0])
        o = OracleMatrix(fmt, c["mode"], kern[mat + "_cols"], kern[mat + "_rows"], kern[mat + "_vals"], n)
        o.inject(c["index"], c["bits"])
        x = kern[mat + "_x"]
        y1 = o.spmv(x)
        ev, fatal = o.events()
        text = "".join(event_lines(ev, fmt))
        assert fatal == (c["exit"] == 1), c
        if fatal:
            assert text == c["stdout"], c
            continue
        y2 = o.spmv(x)
        ev2, _ = o.events()
        assert text + "".join(event_lines(ev2, fmt)) == c["stdout"], c
        assert [format(int(v), "016x") for v in y1.view(np.uint64)] == c["y1"], c
        assert [format(int(v), "016x") for v in y2.view(np.uint64)] == c["y2"], c
        assert [int(v) for v in o.stored_words()[c["index"]]] == c["words_after"][0], c
