"""Pins the oracle's partition builders and build_local_nonlocal against
reference/test/distributed/{matrix,partition}_kernels.cpp, and checks the
product's host-side Partition metadata (C ABI, no GPU needed) against it."""
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "distributed.json")))


def oracle_partition_from_mapping(oracle, mapping, num_parts):
    mapping = np.array(mapping, np.int32)
    n = len(mapping)
    bounds, ids = np.zeros(n + 1, np.int64), np.zeros(max(n, 1), np.int32)
    nr = oracle.ref_partition_build_from_mapping(n, mapping, bounds, ids)
    starts, sizes = np.zeros(max(nr, 1), np.int32), np.zeros(num_parts, np.int32)
    oracle.ref_partition_build_starting_indices(bounds, ids, nr, num_parts, starts, sizes)
    return bounds[:nr + 1].copy(), ids[:nr].copy(), starts[:nr].copy(), sizes, nr


def oracle_build(oracle, rows, cols, vals, part_meta, num_parts, part, col_meta=None):
    bounds, ids, starts, sizes, nr = part_meta
    cbounds, cids, cstarts, _, cnr = part_meta if col_meta is None else col_meta
    rows, cols, vals = np.array(rows, np.int64), np.array(cols, np.int64), np.array(vals, np.float64)
    nnz = len(rows)
    m = max(nnz, 1)
    o = dict(l_rows=np.zeros(m, np.int32), l_cols=np.zeros(m, np.int32), l_vals=np.zeros(m), nl_rows=np.zeros(m, np.int32),
             nl_cols=np.zeros(m, np.int32), nl_vals=np.zeros(m), gather=np.zeros(m, np.int32),
             recv=np.zeros(num_parts, np.int32), n2g=np.zeros(m, np.int64))
    sz = np.zeros(3, np.int64)
    z64, zf = np.zeros(1, np.int64), np.zeros(1)
    oracle.ref_dist_build_local_nonlocal(nnz, rows if nnz else z64, cols if nnz else z64, vals if nnz else zf, bounds, ids,
                                         starts, nr, cbounds, cids, cstarts, cnr, num_parts, part, o["l_rows"], o["l_cols"],
                                         o["l_vals"], o["nl_rows"], o["nl_cols"], o["nl_vals"], o["gather"], o["recv"],
                                         o["n2g"], sz)
    return o, sz


@pytest.mark.parametrize("case", G["build_local_nonlocal"], ids=lambda c: c["name"])
def test_build_local_nonlocal(oracle, case):
    meta = oracle_partition_from_mapping(oracle, case["mapping"], case["num_parts"])
    cmeta = oracle_partition_from_mapping(oracle, case["col_mapping"], case["num_parts"]) if "col_mapping" in case else None
    for part in range(case["num_parts"]):
        o, sz = oracle_build(oracle, case["rows"], case["cols"], case["vals"], meta, case["num_parts"], part, cmeta)
        e = case["local"][part]
        assert list(o["l_rows"][:sz[0]]) == e["rows"] and list(o["l_cols"][:sz[0]]) == e["cols"]
        assert list(o["l_vals"][:sz[0]]) == e["vals"]
        e = case["non_local"][part]
        assert list(o["nl_rows"][:sz[1]]) == e["rows"] and list(o["nl_cols"][:sz[1]]) == e["cols"]
        assert list(o["nl_vals"][:sz[1]]) == e["vals"]
        assert list(o["gather"][:sz[2]]) == case["gather_idxs"][part]
        assert list(o["recv"]) == case["recv_sizes"][part]


@pytest.mark.parametrize("case", G["uniform_ranges"], ids=lambda c: f"{c['num_parts']}x{c['global_size']}")
def test_uniform_partition(oracle, gk, case):
    import gkomi.distributed as gd
    r = np.zeros(case["num_parts"] + 1, np.int64)
    oracle.ref_partition_build_ranges_from_global_size(case["num_parts"], case["global_size"], r)
    assert list(r) == case["ranges"]
    p = gd.Partition.build_from_global_size_uniform(gk, case["num_parts"], case["global_size"])
    assert list(p.range_bounds) == case["ranges"] and list(p.part_ids) == list(range(case["num_parts"]))
    assert list(p.starts) == [0] * case["num_parts"] and list(p.part_sizes) == case["part_sizes"]


def test_partition_from_mapping_matches_oracle(oracle, gk):
    import gkomi.distributed as gd
    for mapping, nparts in (([1, 0, 2, 2, 0, 1, 1, 2], 3), ([1, 2, 0, 0, 2, 1], 3), ([0, 0, 0], 2)):
        b, i, s, z, nr = oracle_partition_from_mapping(oracle, mapping, nparts)
        p = gd.Partition.build_from_mapping(gk, mapping, nparts)
        assert list(p.range_bounds) == list(b) and list(p.part_ids) == list(i)
        assert list(p.starts) == list(s) and list(p.part_sizes) == list(z)


@pytest.mark.parametrize("case", G["ghost_maps"], ids=lambda c: c["name"])
def test_ghost_maps(oracle, case):
    """non_local_to_global: the global columns of the non-local block, sorted by (owning part, column)
    (reference/test/distributed/matrix_kernels.cpp:516-565)"""
    meta = oracle_partition_from_mapping(oracle, case["mapping"], case["num_parts"])
    for part in range(case["num_parts"]):
        o, sz = oracle_build(oracle, case["rows"], case["cols"], case["vals"], meta, case["num_parts"], part)
        assert list(o["n2g"][:sz[2]]) == case["non_local_to_global"][part]


def product_partition(gk, case):
    import gkomi.distributed as gd
    if "mapping" in case:
        return gd.Partition.build_from_mapping(gk, case["mapping"], case["num_parts"])
    if "ranges" in case:
        return gd.Partition.build_from_contiguous(gk, case["ranges"])
    return gd.Partition.build_from_global_size_uniform(gk, *case["uniform"])


@pytest.mark.parametrize("case", G["partitions"], ids=lambda c: c["name"])
def test_partition_builders(oracle, gk, case):
    """reference/test/distributed/partition_kernels.cpp:88-223: the oracle's builders and the product's host-side
    Partition (C ABI, no GPU) against the reference's expected arrays"""
    np_ = case["num_parts"]
    if "mapping" in case:
        bounds, ids, starts, sizes, nr = oracle_partition_from_mapping(oracle, case["mapping"], np_)
        empty = int(np.count_nonzero(sizes == 0))
    else:
        ranges = np.array(case["ranges"], np.int64) if "ranges" in case else np.zeros(np_ + 1, np.int64)
        if "uniform" in case:
            oracle.ref_partition_build_ranges_from_global_size(*case["uniform"], ranges)
        bounds, ids = np.zeros(np_ + 1, np.int64), np.zeros(max(np_, 1), np.int32)
        oracle.ref_partition_build_from_contiguous(np_, ranges, bounds, ids)
        starts, sizes = np.zeros(max(np_, 1), np.int32), np.zeros(max(np_, 1), np.int32)
        empty = oracle.ref_partition_build_starting_indices(bounds, ids, np_, np_, starts, sizes)
        ids, starts, sizes, nr = ids[:np_], starts[:np_], sizes[:np_], np_
    p = product_partition(gk, case)
    for got in ((bounds, ids, starts, sizes, nr, empty),
                (p.range_bounds, p.part_ids, p.starts, p.part_sizes, p.num_ranges, p.num_empty_parts)):
        assert list(got[0][:got[4] + 1]) == case["range_bounds"] and list(got[1][:got[4]]) == case["part_ids"]
        assert list(got[2][:got[4]]) == case["starting_indices"] and list(got[3]) == case["part_sizes"]
        assert got[5] == case["num_empty_parts"]
    assert p.size == case["size"] and p.num_parts == np_


@pytest.mark.parametrize("case", G["partition_properties"], ids=lambda c: c["name"])
def test_partition_connected_ordered(oracle, gk, case):
    """partition_kernels.cpp:226-295"""
    bounds, ids, starts, sizes, nr = oracle_partition_from_mapping(oracle, case["mapping"], case["num_parts"])
    connected = case["num_parts"] - int(np.count_nonzero(sizes == 0)) == nr
    ordered = connected and bool(oracle.ref_partition_has_ordered_parts(ids if nr else np.zeros(1, np.int32), nr))
    p = product_partition(gk, case)
    if "connected" in case:
        assert connected == case["connected"] and p.has_connected_parts() == case["connected"]
    if "ordered" in case:
        assert ordered == case["ordered"] and p.has_ordered_parts() == case["ordered"]


def oracle_vector_build_local(oracle, case, meta, part):
    bounds, ids, starts, sizes, nr = meta
    ncols = case["size"][1]
    local = np.zeros((int(sizes[part]), max(ncols, 1)))
    rows, cols, vals = np.array(case["rows"], np.int64), np.array(case["cols"], np.int64), np.array(case["vals"], np.float64)
    if len(rows):
        oracle.ref_dist_vector_build_local(len(rows), rows, cols, vals, bounds, ids, starts, nr, part, local, local.shape[1])
    return local[:, :ncols]


@pytest.mark.parametrize("case", G["vector_build_local"], ids=lambda c: c["name"])
def test_vector_build_local(oracle, case):
    """reference/test/distributed/vector_kernels.cpp:105-152"""
    meta = oracle_partition_from_mapping(oracle, case["mapping"], case["num_parts"])
    for part in range(case["num_parts"]):
        got = oracle_vector_build_local(oracle, case, meta, part)
        expect = np.array(case["local"][part], np.float64).reshape(got.shape)
        assert np.array_equal(got, expect)
