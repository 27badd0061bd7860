"""Host model of the solver state (no GPU, no reference): Parameter_int / Parameter_real,
BC table, time levels, join tables -- against the layouts the reference documents and the
golden inputs generated together with the reference's outputs."""
import os
import numpy as np
import pytest

from fast_amd import cases, connect, FastC, Internal
import fast_amd.pytree_lite as C
import fast_amd.PyTree as FastS
from common import prepare, zone_arrays

HERE = os.path.dirname(os.path.abspath(__file__))


def _zone(t):
    return Internal.getZones(t)[0]


@pytest.mark.parametrize("scheme,kflu", [('ausmpred', 1), ('senseur', 2), ('roe', 5)])
@pytest.mark.parametrize("model,iflow", [('Euler', 1), ('NSLaminar', 2)])
def test_parameter_int_slots(scheme, kflu, model, iflow):
    """Slots of Fast/FastS/FastS/param_solver.h:77-214 the time step reads."""
    t = cases.tgv((10, 7, 5), scheme, model=model)
    prepare(t)
    p, pr, _ = zone_arrays(_zone(t))
    assert list(p[FastC.NIJK:FastC.NIJK + 5]) == [14, 11, 9, 2, 2]
    assert list(p[FastC.IJKV:FastC.IJKV + 3]) == [10, 7, 5]
    assert p[FastC.ITYPZONE] == 2            # 3-D Cartesian
    assert p[FastC.ITYPCP] == 2              # explicit
    assert p[FastC.KFLUDOM] == kflu and p[FastC.IFLOW] == iflow
    assert p[FastC.NEQ] == 5 and p[FastC.NSSITER] == 3 and p[FastC.RK] == 3
    assert p[FastC.SLOPE] == 2               # 'o3'
    assert p[FastC.NDIMDX] >= 14 * 11 * 9
    assert p[FastC.NEQ_IJ] == 1 and p[FastC.NEQ_K] == 1 and p[FastC.NDIMDX_MTR] == 1
    assert pr[FastC.DTC] > 0 and abs(pr[FastC.GAMMA] - 1.4) < 1e-15


def test_bc_table_layout():
    """Fast/FastC/FastC/PyTree.py:2958-3014: [nb_bc, pt_1..pt_n, then per BC
    (type, idir, window(6), nbdata, ...)], windows from getRange (cell indices, 1-based)."""
    t = cases.tgv((6, 5, 4), 'senseur')
    prepare(t)
    p, _, _ = zone_arrays(_zone(t))
    pt = int(p[FastC.PT_BC])
    assert p[pt] == 6
    seen = []
    for n in range(6):
        b = int(p[pt + 1 + n])
        assert p[b + FastC.BC_TYPE] == FastC.tagBCDict['BCautoperiod'] == 11
        idir = int(p[b + FastC.BC_IDIR]); seen.append(idir)
        w = list(p[b + FastC.BC_FEN:b + FastC.BC_FEN + 6])
        full = [1, 6, 1, 5, 1, 4]
        d = (idir - 1) // 2
        lo = (idir - 1) % 2 == 0
        exp = list(full)
        exp[2 * d] = exp[2 * d + 1] = 1 if lo else full[2 * d + 1]
        assert w == exp, (idir, w)
    assert sorted(seen) == [1, 2, 3, 4, 5, 6]


def test_bc_tags_match_reference_table():
    # Fast/FastC/FastC/PyTree.py:30-82
    for name, tag in [('BCExtrapolate', 0), ('BCFarfield', 1), ('BCWall', 4), ('BCOutflow', 10),
                      ('BCautoperiod', 11), ('BCInflow', 13), ('BCOverlap', 22), ('BCWallModel', 30)]:
        assert FastC.tagBC(name) == tag
    assert FastC.tagBC('BCNoSuchThing') == -1


def test_three_time_levels_are_compact_and_distinct():
    t = cases.tgv(6, 'senseur')
    prepare(t)
    z = _zone(t)
    p, _, _ = zone_arrays(z)
    nd = int(p[FastC.NDIMDX])
    bufs = [FastC.level_buffer(z, s) for s in ('', '_M1', '_P1')]
    for b in bufs:
        assert b.dtype == np.float64 and b.flags['C_CONTIGUOUS'] and b.size == 5 * nd
    assert len({b.ctypes.data for b in bufs}) == 3
    # the fields of the FlowSolution container are views in the level buffer, Density first
    fs = Internal.getNodeFromName1(z, 'FlowSolution#Centers')
    ro = Internal.getNodeFromName1(fs, 'Density')[1]
    T = Internal.getNodeFromName1(fs, 'Temperature')[1]
    assert ro.ctypes.data == bufs[0].ctypes.data
    assert T.ctypes.data == bufs[0].ctypes.data + 4 * nd * 8
    assert np.array_equal(bufs[0], bufs[1]) and np.array_equal(bufs[0], bufs[2])


def test_switch_pointers_rotation():
    """Fast/FastC/FastC/PyTree.py:2292-2315, order 3: M1 <- N <- P1 <- M1."""
    t = cases.tgv(4, 'senseur')
    prepare(t)
    z = _zone(t)
    a = [FastC.level_buffer(z, s).ctypes.data for s in ('', '_M1', '_P1')]
    FastC.switchPointers__([z], 1, 3)
    b = [FastC.level_buffer(z, s).ctypes.data for s in ('', '_M1', '_P1')]
    assert b[1] == a[0] and b[0] == a[2] and b[2] == a[1]
    FastC.switchPointers__([z], 1, 3); FastC.switchPointers__([z], 1, 3)
    assert [FastC.level_buffer(z, s).ctypes.data for s in ('', '_M1', '_P1')] == a


def test_zone_classification_curvilinear_and_k_homogeneous():
    t = cases.tgv(6, 'senseur')
    z = _zone(t)
    g = Internal.getNodeFromName1(z, 'GridCoordinates')
    x = Internal.getNodeFromName1(g, 'CoordinateX')[1]
    x[3, 3, 3] += 0.01        # one displaced node: curvilinear
    m = prepare(t)
    pint = Internal.getNodeFromName2(z, 'Parameter_int')[1]
    assert pint[25] == 0 and list(pint[5:10]) == [1, 11, 121, 2, 2] and pint[43] == 11 ** 3
    assert m[0][0].size == 10 * 11 ** 3
    t = cases.tgv(6, 'senseur')
    z = _zone(t)
    g = Internal.getNodeFromName1(z, 'GridCoordinates')
    x = Internal.getNodeFromName1(g, 'CoordinateX')[1]
    x[3, 3, :] += 0.01        # the same displacement on every k plane: k-homogeneous (type 1)
    m = prepare(t)
    pint = Internal.getNodeFromName2(z, 'Parameter_int')[1]
    # metrics of one node plane: ti, tj with two components, tk with one, no stride along k
    assert pint[25] == 1 and list(pint[5:10]) == [1, 11, 0, 2, 0] and pint[43] == 11 ** 2
    assert pint[37] == 2 and pint[38] == 1 and m[0][0].size == 6 * 11 ** 2


def test_golden_inputs_reproduced_by_host_model():
    """The Parameter_int / Parameter_real / initial field stored with the reference outputs are
    what the host model builds today (the metrics in the file come from the reference)."""
    g = np.load(os.path.join(HERE, 'golden', 'tgv12_senseur_lamin.npz'))
    t = cases.tgv(12, 'senseur', extra_numz={'cache_blocking_I': 100000, 'cache_blocking_J': 100000,
                                              'cache_blocking_K': 100000})
    m = prepare(t)
    p, pr, _ = zone_arrays(_zone(t))
    assert np.array_equal(p, g['pint0'])
    assert np.array_equal(pr, g['preal0'])
    assert np.array_equal(FastC.level_buffer(_zone(t)), g['ro0_0'])
    assert np.array_equal(m[0][0], g['mtr0'])


# ---- join tables ---------------------------------------------------------------------------
def _script_tables(direction, N):
    """The loops of Fast/FastS/test/tgvMpi.py:141-192 (cubic block of N-1 cells)."""
    ni = N - 1 + 4; ninj = ni * ni
    r, d = [], []
    rng = {'imin': range(-1, 1), 'imax': range(N, N + 2)}
    if direction in ('imin', 'imax'):
        for k in range(1, N):
            for j in range(1, N):
                for i in rng[direction]:
                    r.append((i + 1) + (j + 1) * ni + (k + 1) * ninj)
                    d.append(((i + N) if direction == 'imin' else (i - N + 2)) + (j + 1) * ni + (k + 1) * ninj)
    elif direction in ('jmin', 'jmax'):
        for j in (range(-1, 1) if direction == 'jmin' else range(N, N + 2)):
            for k in range(1, N):
                for i in range(1, N):
                    r.append((i + 1) + (j + 1) * ni + (k + 1) * ninj)
                    d.append((i + 1) + ((j + N) if direction == 'jmin' else (j - N + 2)) * ni + (k + 1) * ninj)
    else:
        for k in (range(-1, 1) if direction == 'kmin' else range(N, N + 2)):
            for j in range(1, N):
                for i in range(1, N):
                    r.append((i + 1) + (j + 1) * ni + (k + 1) * ninj)
                    d.append((i + 1) + (j + 1) * ni + ((k + N) if direction == 'kmin' else (k - N + 2)) * ninj)
    return np.array(r, dtype=np.int32), np.array(d, dtype=np.int32)


@pytest.mark.parametrize("face", connect.FACES)
def test_join_tables_equal_reference_script(face):
    r, d = connect.face_points(face, (7, 7, 7))
    rs, ds = _script_tables(face, 8)
    assert np.array_equal(r, rs) and np.array_equal(d, ds)


def test_joins_cover_exactly_the_face_ghost_cells():
    blocks = (2, 1, 3); nv = (4, 5, 3)
    joins = connect.periodic_block_joins(blocks, nv)
    nb = 6
    ni, nj, nk = nv[0] + 4, nv[1] + 4, nv[2] + 4
    for b in range(nb):
        hit = np.zeros(ni * nj * nk, dtype=int)
        for (br, bd, r, d) in joins:
            if br == b:
                hit[r] += 1
                # donors are interior cells
                k = d // (ni * nj); j = (d // ni) % nj; i = d % ni
                assert ((i >= 2) & (i < ni - 2) & (j >= 2) & (j < nj - 2) & (k >= 2) & (k < nk - 2)).all()
        h = hit.reshape(nk, nj, ni)
        assert h.max() == 1
        assert h[2:-2, 2:-2, :2].all() and h[2:-2, 2:-2, -2:].all()       # i ghosts (split)
        assert h[:2, 2:-2, 2:-2].all() and h[-2:, 2:-2, 2:-2].all()       # k ghosts (split)
        assert not h[2:-2, :2, 2:-2].any()                                  # j is periodic by BC
        assert not h[2:-2, 2:-2, 2:-2].any()


def test_connectivity_tree_round_trip():
    blocks = (1, 2, 2)
    t = cases.tgv((6, 8, 8), 'senseur', blocks=blocks)
    names = [z[0] for z in Internal.getZones(t)]
    assert names == ['cart0', 'cart1', 'cart2', 'cart3']
    tc = C.node2Center(t)
    joins = connect.periodic_block_joins(blocks, (6, 4, 4))
    connect.add_joins_to_tc(tc, joins, names)
    back = FastS.joins_from_tc(t, tc)
    # one sub-region per (donor, receiver) pair, its faces concatenated
    pairs = {}
    for (br, bd, r, d) in joins:
        pairs.setdefault((names[br], names[bd]), []).append((r, d))
    assert len(back) == len(pairs)
    for (rn, dn, r, d) in back:
        assert np.array_equal(r, np.concatenate([a for a, _ in pairs[(rn, dn)]]))
        assert np.array_equal(d, np.concatenate([b for _, b in pairs[(rn, dn)]]))
    g = connect.graph_of(blocks, names, [0, 1, 2, 3])
    assert g['procDict'] == {'cart0': 0, 'cart1': 1, 'cart2': 2, 'cart3': 3}
    assert sorted(g['graphID'][0]) == [1, 2] and g['graphID'][0][1] == ['cart1']


def test_split_faces_carry_no_bc():
    t = cases.tgv((6, 8, 8), 'senseur', blocks=(1, 2, 1))
    prepare(t)
    p, _, _ = zone_arrays(_zone(t))
    pt = int(p[FastC.PT_BC])
    dirs = sorted(int(p[int(p[pt + 1 + n]) + 1]) for n in range(int(p[pt])))
    assert dirs == [1, 2, 5, 6]


def test_flat_connectivity_layout_reads_back():
    """connect.flatten_joins lays the joins out so that the reference's reads
    (Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:114-118,220,474-478,646-653,748-758) find them again."""
    import numpy as np
    from fast_amd import connect
    blocks, nv = (1, 2, 2), (4, 3, 5)
    names = ['cart%d' % b for b in range(4)]
    owner = {names[b]: b // 2 for b in range(4)}
    joins = [(names[br], names[bd], r, d) for br, bd, r, d in connect.periodic_block_joins(blocks, nv) if owner[names[bd]] == 1]
    pint, preal = connect.flatten_joins(joins, {'cart2': 0, 'cart3': 1}, owner, 1, {n: int(n[4:]) % 2 for n in names})
    assert pint[0] == 0 and pint[2] == 0 and pint[3] == 0
    shift_graph = pint[2] + pint[3 + pint[2]] + 3
    seen = []
    for ip2p in range(1, pint[1] + 1):
        ech = pint[ip2p + shift_graph]
        dest, nrac, nrac_inst, timelevel = pint[ech:ech + 4]
        assert nrac_inst == 0 and timelevel == 0
        for irac in range(nrac):
            sr = ech + 4 + 2 * timelevel + irac
            assert pint[sr + nrac * 3] < 0
            nod, nor = pint[sr + nrac * 5], pint[sr + nrac * 11 + 1]
            n = pint[sr + nrac * 10 + 1]
            don = pint[pint[sr + nrac * 6]:][:n]; rcv = pint[pint[sr + nrac * 12 + 1]:][:n]
            pos = pint[sr + nrac * 7]; ntype = pint[pos]
            types = pint[pos + 1 + ntype:][:n]
            assert ntype == 1 and pint[pos + 1] == n and (types == 1).all()
            assert (preal[pint[sr + nrac * 8]:][:pint[sr + nrac]] == 1.).all()
            assert pint[sr + nrac * 13 + 1] == 5 and pint[sr + nrac * 14 + 1] == 0
            seen.append((int(dest), int(nor), int(nod), rcv.copy(), don.copy()))
    assert [s[0] for s in seen] == sorted([s[0] for s in seen], key=lambda d: (d != 1, d))     # this rank's block first
    want = [(owner[rn], int(rn[4:]) % 2, int(dn[4:]) % 2, r, d) for rn, dn, r, d in joins]
    assert len(seen) == len(want)
    for s in seen:
        assert any(s[:3] == w[:3] and np.array_equal(s[3], w[3]) and np.array_equal(s[4], w[4]) for w in want)
