"""Match-join ("raccord") tables between the blocks of a split periodic box.

The reference's TGV script builds these tables by hand for a cubic block and a 1 x 2 x 2
split (Fast/FastS/test/tgvMpi.py:80-195): for every face of a block that touches another
block, the two ghost layers of the receiver take the values of the two interior layers of
the donor next to the shared face.  Indices are 0-based, ghost-inclusive and linear
(i + j*ni + k*ni*nj), and the tables live in the connectivity tree under the DONOR zone as a
ZoneSubRegion_t 'ID_<receiver>' with PointList (donor cells) / PointListDonor (receiver cells),
which is the storage K_FASTC::setInterpTransfers consumes
(Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:440-935).

Here the same tables are produced for any (bi,bj,bk) split and any block size; the loop nest
order of the reference script is kept (i fastest inside a j-k sweep for i-faces, and so on) so
that a 1 x 2 x 2 split of a cubic box gives the script's arrays element for element.
"""
import numpy
from . import Internal

FACES = ('imin', 'imax', 'jmin', 'jmax', 'kmin', 'kmax')


def face_points(face, nv, ghost=2):
    """(receiver_pts, donor_pts) of one face of a block with nv = (iv,jv,kv) interior cells whose
    neighbour across that face has the same interior size along the face normal.
    Receiver cells are the ghost layers of `face`; donor cells are the interior layers of the
    neighbour that coincide with them (tgvMpi.py:149-192)."""
    iv, jv, kv = nv
    ni = iv + 2 * ghost; nj = jv + 2 * ghost
    ninj = ni * nj
    d = FACES.index(face) // 2
    hi = FACES.index(face) % 2
    n = nv[d]
    # interior-based index (1..n interior): ghost layers are -1,0 (min) and n+1,n+2 (max)
    g = numpy.arange(-1, 1) if not hi else numpy.arange(n + 1, n + 3)
    gd = g + n if not hi else g - n          # coinciding interior cells of the neighbour
    ri = numpy.arange(1, iv + 1); rj = numpy.arange(1, jv + 1); rk = numpy.arange(1, kv + 1)

    def lin(i, j, k):
        return ((i + ghost - 1) + (j + ghost - 1) * ni + (k + ghost - 1) * ninj).astype(numpy.int32).ravel()

    if d == 0:      # loops k, j, i(ghost)
        K, J, I = numpy.meshgrid(rk, rj, g, indexing='ij')
        _, _, Id = numpy.meshgrid(rk, rj, gd, indexing='ij')
        return lin(I, J, K), lin(Id, J, K)
    if d == 1:      # loops j(ghost), k, i
        J, K, I = numpy.meshgrid(g, rk, ri, indexing='ij')
        Jd, _, _ = numpy.meshgrid(gd, rk, ri, indexing='ij')
        return lin(I, J, K), lin(I, Jd, K)
    K, J, I = numpy.meshgrid(g, rj, ri, indexing='ij')   # loops k(ghost), j, i
    Kd, _, _ = numpy.meshgrid(gd, rj, ri, indexing='ij')
    return lin(I, J, K), lin(I, J, Kd)


def cmesh_wake_join(ni, nj, nwake, ghost=2):
    """(receiver_pts, donor_pts) of the match join that closes the wake cut of a two-dimensional C-mesh with
    ni x nj interior cells whose jmin line is [lower wake | section | upper wake], nwake cells per wake side:
    the ghost cell (i, 1 - r) below the cut coincides with the interior cell (ni + 1 - i, r) of the other side
    (r = 1, 2).  The zone is its own donor.  Two-dimensional zone: no ghost layer along k."""
    nis = ni + 2 * ghost
    rp, dp = [], []
    for r in range(1, ghost + 1):
        for i in list(range(1, nwake + 1)) + list(range(ni - nwake + 1, ni + 1)):
            rp.append((i + ghost - 1) + (1 - r + ghost - 1) * nis)
            dp.append((ni + 1 - i + ghost - 1) + (r + ghost - 1) * nis)
    return numpy.array(rp, dtype=numpy.int32), numpy.array(dp, dtype=numpy.int32)


def block_index(pos, blocks):
    return pos[0] + pos[1] * blocks[0] + pos[2] * blocks[0] * blocks[1]


def block_pos(b, blocks):
    i = b % blocks[0]
    j = (b // blocks[0]) % blocks[1]
    k = b // (blocks[0] * blocks[1])
    return (i, j, k)


def split_faces(blocks):
    """Faces that are joins (the direction is split) rather than periodic BCs."""
    out = []
    for d in range(3):
        if blocks[d] > 1:
            out += [FACES[2 * d], FACES[2 * d + 1]]
    return out


def periodic_block_joins(blocks, nv):
    """All joins of a periodic box split in blocks = (bi,bj,bk) equal blocks of nv interior
    cells: list of (receiver_block, donor_block, receiver_pts, donor_pts), ordered by donor
    block then face, the order in which a donor-side connectivity tree is walked."""
    out = []
    nb = blocks[0] * blocks[1] * blocks[2]
    for bd in range(nb):
        pos = block_pos(bd, blocks)
        for face in split_faces(blocks):
            d = FACES.index(face) // 2
            hi = FACES.index(face) % 2
            # `face` names the RECEIVER's face (the script's convention): its ghost layers there
            # are filled by the block sitting across that face, so seen from the donor the
            # receiver is the neighbour on the opposite side
            npos = list(pos)
            npos[d] = (pos[d] + (-1 if hi else 1)) % blocks[d]
            br = block_index(npos, blocks)
            rpts, dpts = face_points(face, nv)
            out.append((br, bd, rpts, dpts))
    return out


def add_joins_to_tc(tc, joins, zone_names, local=None):
    """Store joins in the connectivity tree the way the reference expects them
    (tgvMpi.py:132-195): one ZoneSubRegion_t per (donor, receiver) pair, several faces
    concatenated.  `local` restricts to donor zones present in tc."""
    by_pair = {}
    for (br, bd, rpts, dpts) in joins:
        by_pair.setdefault((bd, br), []).append((rpts, dpts))
    zones = {z[0]: z for z in Internal.getZones(tc)}
    for (bd, br), lst in by_pair.items():
        dn = zone_names[bd]; rn = zone_names[br]
        if dn not in zones:
            if local is None:
                raise ValueError("add_joins_to_tc: donor zone %s is not in tc" % dn)
            continue
        z = zones[dn]
        rp = numpy.concatenate([a for a, _ in lst]).astype(Internal.E_NpyInt)
        dp = numpy.concatenate([b for _, b in lst]).astype(Internal.E_NpyInt)
        sr = Internal.createUniqueChild(z, 'ID_' + rn, 'ZoneSubRegion_t', rn)
        Internal.createUniqueChild(sr, 'ZoneRole', 'DataArray_t', 'Donor')
        Internal.createUniqueChild(sr, 'GridLocation', 'GridLocation_t', 'CellCenter')
        Internal.createUniqueChild(sr, 'InterpolantsDonor', 'DataArray_t', numpy.ones(rp.size))
        Internal.createUniqueChild(sr, 'InterpolantsType', 'DataArray_t',
                                   numpy.ones(rp.size, dtype=Internal.E_NpyInt))
        Internal.createUniqueChild(sr, 'PointListDonor', 'DataArray_t', rp)
        Internal.createUniqueChild(sr, 'PointList', 'DataArray_t', dp)
    return tc


def graph_of(blocks, zone_names, owner):
    """The 'graph' dictionary the reference's Mpi entry points take (tgvMpi.py:198-256):
    graphID[rank] = {peer_rank: [zone names on the peer that receive from rank]}, procDict[zone]
    = owning rank."""
    nb = len(zone_names)
    procDict = {zone_names[b]: owner[b] for b in range(nb)}
    nranks = max(owner) + 1
    graphID = {r: {} for r in range(nranks)}
    for bd in range(nb):
        pos = block_pos(bd, blocks)
        for face in split_faces(blocks):
            d = FACES.index(face) // 2
            hi = FACES.index(face) % 2
            npos = list(pos); npos[d] = (pos[d] + (-1 if hi else 1)) % blocks[d]
            br = block_index(npos, blocks)
            if owner[br] != owner[bd]:
                lst = graphID[owner[bd]].setdefault(owner[br], [])
                if zone_names[br] not in lst:
                    lst.append(zone_names[br])
    procList = [[zone_names[b] for b in range(nb) if owner[b] == r] for r in range(nranks)]
    return {'graphID': graphID, 'graphIBCD': {}, 'procDict': procDict, 'procList': procList}


def flatten_joins(joins, zone_index, owner, rank=0, index_on_owner=None):
    """tc/Parameter_int and tc/Parameter_real of a set of match joins, the flat layout the reference's transfer
    routines read (Fast/FastC/FastC/Com/setInterpTransfersFast.cpp:114-118,220,469-478,522-528,646-653,748-758;
    in the reference Cassiopee's miseAPlatDonorTree__ produces it from the ZoneSubRegion_t nodes of tc):

        [0] 0 (not initialised)   [1] number of point-to-point blocks   [2] 0 (no IBC graph)   [3] 0 (no ID graph)
        [4 ...] pointer of every block; a block = [destination rank, nrac, 0, 0] + 16 arrays of nrac entries (one
        spare entry after the tenth) + 27 entries per rac (local time stepping levels, unused) + per rac the
        donor points, the receiver points and the type table [1, npts, 1 x npts]; coefficients (1.) in Parameter_real.

    joins = [(receiver zone name, donor zone name, receiver points, donor points)] with the donor on this rank;
    zone_index = {name: index on this rank}; owner = {name: rank}; index_on_owner = {name: index on its own rank}
    (default: zone_index).  One block per destination rank, this rank first."""
    if index_on_owner is None:
        index_on_owner = zone_index
    by_dest = {}
    for j in joins:
        by_dest.setdefault(owner.get(j[0], rank) if owner else rank, []).append(j)
    dests = sorted(by_dest, key=lambda d: (d != rank, d))
    np2p = len(dests)
    pint = [0, np2p, 0, 0] + [0] * np2p
    preal = []
    for n, d in enumerate(dests):
        racs = by_dest[d]
        nrac = len(racs)
        ech = len(pint)
        pint[4 + n] = ech
        pint += [d, nrac, 0, 0]
        base = len(pint)                       # shift_rac of irac 0
        pint += [0] * (16 * nrac + 1 + 27 * nrac)
        for irac, (rn, dn, rpts, dpts) in enumerate(racs):
            rp = numpy.asarray(rpts, dtype=numpy.int32).ravel()
            dp = numpy.asarray(dpts, dtype=numpy.int32).ravel()
            if rp.size != dp.size:
                raise ValueError("flatten_joins: join %s <- %s: lists differ in length" % (rn, dn))
            sr = base + irac
            pint[sr] = int(dp.size)                                  # donor points
            pint[sr + nrac] = int(rp.size)                           # NbInterpD: coefficients of the rac
            pint[sr + 3 * nrac] = -1                                 # ibcType < 0: interpolation data
            pint[sr + 5 * nrac] = int(zone_index[dn])
            pint[sr + 6 * nrac] = len(pint); pint += dp.tolist()
            pint[sr + 7 * nrac] = len(pint); pint += [1, int(rp.size)] + [1] * int(rp.size)
            pint[sr + 8 * nrac] = len(preal); preal += [1.] * int(rp.size)
            pint[sr + 10 * nrac + 1] = int(rp.size)
            pint[sr + 11 * nrac + 1] = int(index_on_owner[rn])
            pint[sr + 12 * nrac + 1] = len(pint); pint += rp.tolist()
            pint[sr + 13 * nrac + 1] = 5
            pint[sr + 14 * nrac + 1] = 0
    return numpy.array(pint, dtype=numpy.int32), numpy.array(preal if preal else [0.], dtype=numpy.float64)


def add_flat_to_tc(tc, pint, preal):
    """Store the flat arrays in the connectivity tree under the names the reference uses (tc/Parameter_int,
    tc/Parameter_real, read back at Fast/FastS/FastS/PyTree.py:107,488)."""
    Internal.createUniqueChild(tc, 'Parameter_int', 'DataArray_t', pint)
    Internal.createUniqueChild(tc, 'Parameter_real', 'DataArray_t', preal)
    return tc
