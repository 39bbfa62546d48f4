#!/usr/bin/env python3
"""Generate tests/golden/goldens.json from the reference's own functional tests.

Run in the dev container only (needs /root/reference):
    python tests/golden/make_golden.py

For every case we record (a) a small *spec* of the table the reference test builds
(the SQL INSERTs restated as generator segments, see tests/golden_util.py:materialize)
and the query, and (b) the expected rows parsed verbatim from the reference's
`.reference` file.  Float values are kept as the strings ClickHouse printed (shortest
round-trip repr of a Float32), so a bit-exact comparison is `np.float32(s) == value`.

Source directory: /root/reference/tests/queries/2_vector_search/
"""
import json
import os
import re

REF = "/root/reference/tests/queries/2_vector_search"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "goldens.json")


def ref_lines(name):
    with open(os.path.join(REF, name + ".reference")) as f:
        return f.read().split("\n")


def rows3(lines):
    """'id\\t[vec]\\tdist' lines -> (ids, dists-as-strings)"""
    ids, ds = [], []
    for ln in lines:
        p = ln.split("\t")
        ids.append(int(float(p[0])))
        ds.append(p[-1])
    return ids, ds


def rows2(lines):
    ids, ds = [], []
    for ln in lines:
        p = ln.split("\t")
        ids.append(int(float(p[0])))
        ds.append(p[1])
    return ids, ds


def parse_values(sql, table):
    """(id, [a,b,c], 'text'...) tuples of the first INSERT INTO <table> VALUES statement."""
    m = re.search(r"INSERT INTO %s VALUES (.*?);\n" % re.escape(table), sql, re.S)
    body = m.group(1)
    out = []
    for t in re.finditer(r"\(\s*(\d+)\s*,\s*\[([^\]]*)\]\s*,\s*\[?((?:'(?:[^']|'')*'\s*,?\s*)+)\]?\)", body):
        texts = [s.replace("''", "'") for s in re.findall(r"'((?:[^']|'')*)'", t.group(3))]
        out.append({"id": int(t.group(1)), "vector": [float(v) for v in t.group(2).split(",")], "texts": texts})
    return out


def main():
    g = {}
    nnn100 = [{"kind": "nnn", "start": 0, "count": 100, "dim": 3}]

    # 00001: FLAT index, L2, top-10  (helpers/00000_prepare_index.sh)
    ids, ds = rows3(ref_lines("00001_mqvs_distance")[:10])
    g["00001_flat_l2"] = {"source": "00001_mqvs_distance.sh/.reference", "base": nnn100, "metric": "L2",
                          "queries": [[0.1, 0.1, 0.1]], "k": 10, "ids": [ids], "dists": [ds]}

    # 00002: batch_distance, 3 queries, 2 parts (rows 0-49 / 50-99), L2 then IP DESC, LIMIT 10 BY query
    ln = ref_lines("00002_mqvs_batch_distance")
    for tag, start, metric in (("l2", 2, "L2"), ("ip", 33, "IP")):
        blk = ln[start:start + 30]
        idl, dl = [], []
        for q in range(3):
            rows = blk[q * 10:(q + 1) * 10]
            idl.append([int(float(r.split("\t")[0])) for r in rows])
            dl.append([re.match(r"\((\d+),(.*)\)", r.split("\t")[2]).group(2) for r in rows])
            assert all(int(re.match(r"\((\d+),", r.split("\t")[2]).group(1)) == q for r in rows)
        g["00002_batch_" + tag] = {"source": "00002_mqvs_batch_distance.sh/.reference",
                                   "parts": [[{"kind": "nnn", "start": 0, "count": 50, "dim": 3}],
                                             [{"kind": "nnn", "start": 50, "count": 50, "dim": 3}]],
                                   "metric": metric, "queries": [[0.1] * 3, [0.2] * 3, [50.1] * 3], "k": 10,
                                   "ids": idl, "dists": dl}

    # 00003: prewhere id < 10 or id > 60, ORDER BY (d, id) LIMIT 20
    ids, ds = rows3(ref_lines("00003_mqvs_distance_with_prewhere")[:20])
    g["00003_prewhere"] = {"source": "00003_mqvs_distance_with_prewhere.sh/.reference", "base": nnn100,
                           "metric": "L2", "queries": [[1.0, 1.0, 1.0]], "k": 20,
                           "filter": "id < 10 or id > 60", "ids": [ids], "dists": [ds]}

    # 00008: empty vectors, FLAT index, ORDER BY (dist, id) LIMIT 10 (ties 9/31, 8/32 ... ordered by id)
    ids, ds = rows3(ref_lines("00008_mqvs_empty_vector")[10:20])
    g["00008_empty_vectors"] = {"source": "00008_mqvs_empty_vector.sh/.reference (FLAT block)",
                                "base": [{"kind": "nnn", "start": 0, "count": 10, "dim": 3},
                                         {"kind": "empty", "start": 10, "count": 20, "dim": 3},
                                         {"kind": "nnn", "start": 30, "count": 400, "dim": 3}],
                                "metric": "L2", "queries": [[20.0, 20.0, 20.0]], "k": 10, "ids": [ids], "dists": [ds]}

    # 00009-00012: brute force over helpers/00000_prepare_index_2.sh (index_granularity=128, empty rows 10..29)
    bf_base = [{"kind": "nnn", "start": 0, "count": 10, "dim": 3}, {"kind": "empty", "start": 10, "count": 20, "dim": 3},
               {"kind": "nnn", "start": 30, "count": 10000, "dim": 3}]
    for name, q, flt in (("00012_mqvs_brute_force_search", 10020.1, None),
                         ("00009_mqvs_brute_force_search_prewhere_0", 10020.1,
                          "id > 5000 or id == 9 or id == 31 or id == 999 or id == 1"),
                         ("00010_mqvs_brute_force_search_prewhere_1", 10020.1, "id < 100 or id > 10000"),
                         ("00011_mqvs_brute_force_search_where", 10020.0,
                          "id < 50 or id == 51 or id == 55 or id == 99 or id == 100 or id == 9999")):
        lines = [l for l in ref_lines(name) if l.strip()]
        ids, ds = rows3(lines)
        g[name[:5] + "_brute_force" + ("" if flt is None else "_filter")] = {
            "source": name + ".sh/.reference", "base": bf_base, "index_granularity": 128, "metric": "L2",
            "queries": [[q, q, q]], "k": 100, "filter": flt, "ids": [ids], "dists": [ds]}

    # 00014: cosine brute force d=3; cosine via IVFFLAT/HNSW d=4 (same numbers in both references)
    ids, ds = rows2(ref_lines("00014_mqvs_distance_cosine_bruteforce")[:5])
    g["00014_cosine_bruteforce"] = {"source": "00014_mqvs_distance_cosine_bruteforce.sql/.reference",
                                    "base": [{"kind": "n_n3_n1", "start": 0, "count": 1000, "dim": 3}],
                                    "metric": "Cosine", "queries": [[8.0, 11.0, 9.0]], "k": 5, "ids": [ids], "dists": [ds]}
    a = ref_lines("00014_mqvs_distance_cosine_ivfflat")[:10]
    assert a == ref_lines("00014_mqvs_distance_cosine_hnsw")[:10]
    ids, ds = rows3(a)
    g["00014_cosine_d4_index"] = {"source": "00014_mqvs_distance_cosine_{ivfflat,hnsw}.sh/.reference + helpers/00000_prepare_index_cosine.sh",
                                  "base": [{"kind": "cosine4", "start": 2, "count": 3998, "dim": 4}], "metric": "Cosine",
                                  "queries": [[0.5, 0.5, 0.5, 0.5]], "k": 10, "nprobe": 32, "ids": [ids], "dists": [ds]}

    # 00028: 768-d, 1000 rows, MSTG index (exact on this size): L2, cosine, cosine + filter, cosine after LWD of ids 0,2
    ln = ref_lines("00028_mqvs_index_mstg_build_search")
    q768 = [round(0.01 * (i + 1), 2) for i in range(768)]
    base768 = [{"kind": "mstg768", "start": 0, "count": 1000, "dim": 768}]
    for tag, lo, metric, flt, deleted in (("l2", 4, "L2", None, []), ("cosine", 13, "Cosine", None, []),
                                           ("cosine_where", 18, "Cosine", "id != 0", []),
                                           ("cosine_lwd", 23, "Cosine", None, [0, 2])):
        ids, ds = rows2(ln[lo:lo + 5])
        g["00028_768_" + tag] = {"source": "00028_mqvs_index_mstg_build_search.sql/.reference", "base": base768,
                                 "metric": metric, "queries": [q768], "k": 5, "filter": flt, "deleted": deleted,
                                 "ids": [ids], "dists": [ds]}

    # 00040 / 00041: BM25 text search + hybrid fusion
    with open(os.path.join(REF, "00040_mqvs_hybrid_search.sql")) as f:
        sql = f.read()
    docs = parse_values(sql, "t_vector_invert")
    assert len(docs) == 20
    ln = ref_lines("00040_mqvs_hybrid_search")

    def sect(title, n):
        i = ln.index(title)
        return rows2(ln[i + 1:i + 1 + n])

    g["00040_hybrid"] = {
        "source": "00040_mqvs_hybrid_search.sql/.reference", "docs": docs, "vec_query": [1.0, 1.0, 1.0],
        "text_query": "Ancient", "limit": 5,
        "text_search": sect("text search", 2), "text_search_where_id_lt_10": sect("text search with WHERE clause", 1),
        "rsf": sect("hybrid search with relative score fusion", 5), "rrf": sect("hybrid search with rank fusion", 5),
        "rsf_where_id_lt_10": sect("hybrid search rsf with WHERE clause", 5)}
    arr = parse_values(sql, "t_vector_invert_array")
    assert len(arr) == 10
    g["00040_text_array"] = {"source": "00040_mqvs_hybrid_search.sql/.reference ('text search on Array')",
                             "docs": arr, "text_query": "Military Strategy", "limit": 5,
                             "text_search": sect("text search on Array", 4)}
    multi = parse_values(sql, "t_vector_invert_multi")
    assert len(multi) == 20
    g["00040_hybrid_doc2"] = {"source": "00040_mqvs_hybrid_search.sql/.reference ('hybridsearch on doc2')",
                              "docs": [{"id": d["id"], "vector": d["vector"], "texts": [d["texts"][1]]} for d in multi],
                              "vec_query": [1.0, 1.0, 1.0], "text_query": "cultural", "limit": 5,
                              "rsf": sect("hybridsearch on doc2", 5)}

    ln41 = ref_lines("00041_mqvs_text_search_multiple_parts")

    def sect41(title, n):
        i = ln41.index(title)
        return rows2(ln41[i + 1:i + 1 + n])

    g["00041_two_parts"] = {
        "source": "00041_mqvs_text_search_multiple_parts.sql/.reference", "docs": docs, "part_sizes": [10, 10],
        "vec_query": [1.0, 1.0, 1.0], "text_query": "Ancient", "limit": 5,
        "text_search_2parts": sect41("Text search result with 2 parts", 2),
        "rsf_2parts": sect41("Hybrid search RSF result with 2 parts", 5),
        "text_search_1part": sect41("Text search result with 1 part after optimize final", 2),
        "rsf_1part": sect41("Hybrid search RSF result with 1 part after optimize final", 5)}

    # 00016 / 00032: lightweight deletes (delete bitmap over an index that is exact on these sizes; small marks in 00032)
    ids, ds = rows3(ref_lines("00016_mqvs_lightweight_delete_with_vector")[2:12])
    g["00016_lwd"] = {"source": "00016_mqvs_lightweight_delete_with_vector.sql/.reference",
                      "base": [{"kind": "nnn", "start": 0, "count": 2100, "dim": 3}], "index_granularity": 1024,
                      "metric": "L2", "queries": [[0.1, 0.1, 0.1]], "k": 10, "deleted": [2], "ids": [ids], "dists": [ds]}
    ids, ds = rows3(ref_lines("00032_mqvs_lightweight_delete_small_ranges")[2:12])
    g["00032_lwd_small_ranges"] = {"source": "00032_mqvs_lightweight_delete_small_ranges.sql/.reference",
                                   "base": [{"kind": "nnn", "start": 0, "count": 100, "dim": 3}], "index_granularity": 3,
                                   "metric": "L2", "queries": [[1.0, 1.0, 1.0]], "k": 10, "deleted": [2, 3, 8],
                                   "ids": [ids], "dists": [ds]}

    # 00038: binary vectors FixedString(4), rows char(n, n, n, n) for n in 0..1023 (bytes n % 256), brute force Hamming /
    # Jaccard: single query, batch of 3 (LIMIT 10 BY query), WHERE id > 100 and id < 120; after LWD of id < 200 (Hamming)
    ln = ref_lines("00038_mqvs_binary_vector_feature")

    def sect38(title, n, occurrence=0):
        idx = [i for i, l in enumerate(ln) if l == title][occurrence]
        return ln[idx + 1:idx + 1 + n]

    def batch38(lines):
        idl, dl = [[], [], []], [[], [], []]
        for r in lines:
            i, t = r.split("\t")
            m = re.match(r"\((\d+),(.*)\)", t)
            idl[int(m.group(1))].append(int(i))
            dl[int(m.group(1))].append(m.group(2))
        return idl, dl

    q1 = [100, 101, 102, 103]
    qb = [[0x55, 0x55, 0x55, 0x55], [0, 255, 1, 254], [255, 255, 255, 255]]
    for metric, tag in (("Hamming", "hamming"), ("Jaccard", "jaccard")):
        ids, ds = rows2(sect38("-- Brute Force (%s)" % metric, 20))
        bi, bd = batch38(sect38("-- Batch distance (%s)" % metric, 30))
        fi, fd = rows2(sect38("-- Search with filter (%s)" % metric, 19))
        g["00038_binary_" + tag] = {"source": "00038_mqvs_binary_vector_feature.sql/.reference", "rows": 1024, "nbytes": 4,
                                    "metric": metric, "query": q1, "k": 20, "ids": ids, "dists": ds,
                                    "batch_queries": qb, "batch_k": 10, "batch_ids": bi, "batch_dists": bd,
                                    "filter": "id > 100 and id < 120", "filter_ids": fi, "filter_dists": fd}
    ids, ds = rows2(sect38("-- LWD", 10))
    g["00038_binary_hamming"]["lwd_deleted_below"] = 200
    g["00038_binary_hamming"]["lwd_ids"], g["00038_binary_hamming"]["lwd_dists"] = ids, ds

    with open(OUT, "w") as f:
        json.dump(g, f, indent=1)
    print("wrote", OUT, "cases:", len(g))


if __name__ == "__main__":
    main()
