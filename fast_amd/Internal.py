"""Minimal CGNS/Python "pyTree" node helpers.

A pyTree node is the 4-list ``[name, value, children, cgns_type]`` that Cassiopee's
``Converter.Internal`` manipulates (used throughout the reference API, e.g.
Fast/FastS/FastS/PyTree.py:59-63).  When Cassiopee is importable the host code uses
it; this module is the stand-alone subset needed by the FastS time-step path so the
package works on a machine that only has numpy.  Function names and argument
meaning follow ``Converter.Internal``.
"""
import numpy

E_NpyInt = numpy.int32

__FlowSolutionNodes__ = 'FlowSolution'
__FlowSolutionCenters__ = 'FlowSolution#Centers'
__GridCoordinates__ = 'GridCoordinates'


def _as_value(value):
    """Encode a python value the way CGNS/Python trees store it."""
    if value is None:
        return None
    if isinstance(value, numpy.ndarray):
        return value
    if isinstance(value, str):
        return numpy.frombuffer(value.encode('ascii'), dtype='S1').copy()
    if isinstance(value, (bool, numpy.bool_)):
        return numpy.array([int(value)], dtype=E_NpyInt)
    if isinstance(value, (int, numpy.integer)):
        return numpy.array([value], dtype=E_NpyInt)
    if isinstance(value, (float, numpy.floating)):
        return numpy.array([value], dtype=numpy.float64)
    if isinstance(value, (list, tuple)):
        a = numpy.array(value)
        if a.dtype.kind in 'iu':
            return a.astype(E_NpyInt)
        if a.dtype.kind == 'f':
            return a.astype(numpy.float64)
        return a
    return value


def getValue(node):
    """Return the node value as a python str / scalar / numpy array."""
    v = node[1]
    if v is None:
        return None
    if isinstance(v, numpy.ndarray):
        if v.dtype.kind == 'S':
            return v.tobytes().decode('ascii')
        if v.size == 1:
            return v.reshape(-1)[0].item()
        return v
    return v


def setValue(node, value):
    node[1] = _as_value(value)


def createNode(name, ntype, value=None, children=None, parent=None):
    node = [name, _as_value(value), children if children is not None else [], ntype]
    if parent is not None:
        parent[2].append(node)
    return node


def createChild(parent, name, ntype, value=None, children=None, pos=-1):
    node = createNode(name, ntype, value, children)
    if pos == -1:
        parent[2].append(node)
    else:
        parent[2].insert(pos, node)
    return node


def createUniqueChild(parent, name, ntype, value=None, children=None, pos=-1):
    for c in parent[2]:
        if c[0] == name:
            c[3] = ntype
            c[1] = _as_value(value)
            if children is not None:
                c[2] = children
            return c
    return createChild(parent, name, ntype, value, children, pos)


def addChild(parent, node, pos=-1):
    if pos == -1:
        parent[2].append(node)
    else:
        parent[2].insert(pos, node)
    return node


def getNodeFromName1(node, name):
    for c in node[2]:
        if c[0] == name:
            return c
    return None


def getNodeFromName2(node, name):
    for c in node[2]:
        if c[0] == name:
            return c
    for c in node[2]:
        for d in c[2]:
            if d[0] == name:
                return d
    return None


def getNodeFromName(node, name):
    if node[0] == name:
        return node
    for c in node[2]:
        r = getNodeFromName(c, name)
        if r is not None:
            return r
    return None


def getNodesFromName1(node, name):
    return [c for c in node[2] if c[0] == name]


def getNodesFromName(node, name):
    out = []
    if node[0] == name:
        out.append(node)
    for c in node[2]:
        out += getNodesFromName(c, name)
    return out


def getNodeFromType1(node, ntype):
    for c in node[2]:
        if c[3] == ntype:
            return c
    return None


def getNodesFromType1(node, ntype):
    return [c for c in node[2] if c[3] == ntype]


def getNodesFromType2(node, ntype):
    out = [c for c in node[2] if c[3] == ntype]
    for c in node[2]:
        out += [d for d in c[2] if d[3] == ntype]
    return out


def getNodesFromType3(node, ntype):
    out = getNodesFromType2(node, ntype)
    for c in node[2]:
        for d in c[2]:
            out += [e for e in d[2] if e[3] == ntype]
    return out


def getNodesFromType(node, ntype):
    out = []
    if node[3] == ntype:
        out.append(node)
    for c in node[2]:
        out += getNodesFromType(c, ntype)
    return out


def getBases(t):
    if t[3] == 'CGNSBase_t':
        return [t]
    return getNodesFromType1(t, 'CGNSBase_t')


def getZones(t):
    if t[3] == 'Zone_t':
        return [t]
    if t[3] == 'CGNSBase_t':
        return getNodesFromType1(t, 'Zone_t')
    if t[3] == 'CGNSTree_t':
        out = []
        for b in getNodesFromType1(t, 'CGNSBase_t'):
            out += getNodesFromType1(b, 'Zone_t')
        return out
    if isinstance(t, list) and len(t) > 0 and isinstance(t[0], list):
        out = []
        for z in t:
            out += getZones(z)
        return out
    return []


def getZoneDim(z):
    """('Structured', ni, nj, nk, celldim) in *node* counts, like Converter.Internal."""
    dims = z[1]
    ztype = getNodeFromName1(z, 'ZoneType')
    if ztype is not None and getValue(ztype) != 'Structured':
        return ['Unstructured', int(dims[0][0]), int(dims[0][1]), 'UNKNOWN', 3]
    nd = dims.shape[0]
    ni = int(dims[0][0])
    nj = int(dims[1][0]) if nd > 1 else 1
    nk = int(dims[2][0]) if nd > 2 else 1
    return ['Structured', ni, nj, nk, nd]


def rmNodesByName(t, name):
    t[2] = [c for c in t[2] if c[0] != name]
    for c in t[2]:
        rmNodesByName(c, name)
    return t


def _rmNodesByName(t, name):
    rmNodesByName(t, name)
    return None


def copyRef(node):
    """Copy of the tree structure sharing the numpy values."""
    return [node[0], node[1], [copyRef(c) for c in node[2]], node[3]]


def copyTree(node):
    v = node[1].copy() if isinstance(node[1], numpy.ndarray) else node[1]
    return [node[0], v, [copyTree(c) for c in node[2]], node[3]]


def newCGNSTree():
    t = createNode('CGNSTree', 'CGNSTree_t')
    createChild(t, 'CGNSLibraryVersion', 'CGNSLibraryVersion_t',
                numpy.array([3.1], dtype=numpy.float32))
    return t
