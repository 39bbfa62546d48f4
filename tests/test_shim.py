"""The compiled host shim (shim/HostShim.cpp: namespace Search + the faiss brute-force calls on top of libmsvs.so)
driven by shim/test_shim.cpp the way the reference's host code drives the absent library; results == the CPU oracle."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import oracle as o

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "shim", "_build", "test_shim")


def test_shim_is_built_against_the_stub_headers():
    assert os.path.exists(EXE) and os.path.exists(os.path.join(ROOT, "shim", "_build", "libmsvs_shim.so")), \
        "run `python -c 'import __graft_entry__ as g; g.build()'`"
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", os.path.join(ROOT, "shim", "_build", "libmsvs_shim.so")],
                                  text=True)
    for sym in ("faiss::knn_L2sqr", "faiss::knn_inner_product", "faiss::hammings_knn_mc", "jaccard_knn",
                "Search::createVectorIndex", "Search::getMetricType",
                # what the DDL and the search-argument checks link against (parseVSParameters.cpp:78, VIDescriptions.cpp:41,133,172)
                "Search::MYSCALE_VALID_INDEX_PARAMETER", "Search::getDefaultIndexType"):
        assert sym in out, sym
    # both instantiations the host links against (VICommon.h:142-143): FloatVector = (Search::DataType)0, BinaryVector = 1
    made = [ln for ln in out.split("\n") if "Search::createVectorIndex<" in ln]
    assert any("(Search::DataType)0" in ln for ln in made) and any("(Search::DataType)1" in ln for ln in made), made


def test_forwarding_build_compiles():
    """-DMSVS_SEARCH_FORWARD_SUFFIX: every index type libmsvs does not serve (HNSW*, IVFPQ, IVFSQ, SCANN) goes to the original
    library's factory under its build-time name, the parameter table and the default type come from the original (header of
    shim/HostShim.cpp) -- same device as TextShim.cpp's forwarding namespace."""
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-Istubs",
                           "-DMSVS_SEARCH_FORWARD_SUFFIX=Original", "HostShim.cpp"], cwd=os.path.join(ROOT, "shim"))
    pre = subprocess.check_output(["g++", "-std=c++17", "-E", "-P", "-Istubs", "-DMSVS_SEARCH_FORWARD_SUFFIX=Original", "HostShim.cpp"],
                                  cwd=os.path.join(ROOT, "shim"), text=True)
    assert "createVectorIndexOriginal<IS, OS, Bitmap, T>(name, type" in pre and "getDefaultIndexTypeOriginal(search_type)" in pre
    # (the table is taken from the original's object through a function that falls back to this library's own when the original's
    # initialiser has not run yet: static linking)
    assert "theirs = MYSCALE_VALID_INDEX_PARAMETEROriginal" in pre and "MYSCALE_VALID_INDEX_PARAMETER = forwarded_parameter_table()" in pre


def test_parameter_table_is_the_json_the_host_parses():
    """Search::MYSCALE_VALID_INDEX_PARAMETER as VIDescriptions.cpp:172-330 / parseVSParameters.cpp:78-222 read it: index type (upper
    case) -> parameter -> {type, case_sensitive, range, candidates}; it names what msvs_index_create / msvs_index_search parse and
    the arguments the reference's functional tests pass to the served types."""
    import json
    import re
    src = open(os.path.join(ROOT, "shim", "HostShim.cpp")).read()
    table = json.loads(re.search(r'R"JSON\((.*?)\)JSON"', src, re.S).group(1))
    assert set(table) == {"FLAT", "IVFFLAT", "MSTG", "BINARYFLAT", "BINARYMSTG"}
    for typ, params in table.items():
        assert typ == typ.upper() and "metric_type" in params
        for name, spec in params.items():
            assert set(spec) == {"type", "case_sensitive", "range", "candidates"} and spec["type"] in ("int", "float", "string"), (typ, name)
            assert len(spec["range"]) in (0, 2) and isinstance(spec["candidates"], list)
    assert {"ncentroids", "nprobe", "metric"} <= set(table["IVFFLAT"])  # 00005: TYPE IVFFLAT('metric=IP', 'ncentroids=5000'), distance('nprobe = 8')
    assert {"alpha", "disk_mode"} <= set(table["MSTG"])  # 00028: TYPE mstg('metric_type=Cosine', 'disk_mode=1'), distance('alpha=4.2')
    assert set(table["BINARYFLAT"]["metric_type"]["candidates"]) == {"Hamming", "Jaccard"}


@pytest.mark.gpu
@pytest.mark.parametrize("metric,typ", [("L2", "IVFFLAT"), ("cosine", "MSTG"), ("IP", "FLAT")])
def test_shim_end_to_end_matches_oracle(metric, typ):
    rng = np.random.default_rng(len(metric) * 10 + len(typ))
    n, d, nq, k, nlist = 6000, 40, 7, 10, 12
    x = rng.standard_normal((n, d), dtype=np.float32)
    q = rng.standard_normal((nq, d), dtype=np.float32)
    alive = rng.random(n) < 0.5
    nb_rows, nb_bytes, nb_q = 3000, 24, 4
    bx = rng.integers(0, 256, (nb_rows, nb_bytes), dtype=np.uint8)
    bq = rng.integers(0, 256, (nb_q, nb_bytes), dtype=np.uint8)
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "meta.txt"), "w") as f:
            f.write("%d %d %d %d %d %s %s %d %d %d\n" % (n, d, nq, k, nlist, metric, typ, nb_rows, nb_bytes, nb_q))
        balive = rng.random(nb_rows) < 0.4
        for name, a in (("x", x), ("q", q), ("alive", alive.astype(np.uint8)), ("bx", bx), ("bq", bq), ("balive", balive.astype(np.uint8))):
            a.tofile(os.path.join(td, name + ".bin"))
        r = subprocess.run([EXE, td], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        rd = lambda name, dt: np.fromfile(os.path.join(td, name + ".bin"), dt)
        # both cancellation scenarios left with the host's ABORTED code; the table is JSON; TYPE DEFAULT resolves to served types
        assert "cancel_scenario_0 code 236" in r.stdout and "cancel_scenario_1 code 236" in r.stdout, r.stdout
        import json
        assert "IVFFLAT" in json.load(open(os.path.join(td, "param_table.json")))
        assert open(os.path.join(td, "default_types.txt")).read().split() == ["MSTG", "BinaryMSTG"]
        files = open(os.path.join(td, "files.txt")).read().split("\n")
        assert files[0].startswith("part/v1-data_bin.vidx3 ") and files[1].startswith("part/v1-id_list.vidx3 ")
        assert files[2].startswith("version msvs-")
        om = {"L2": o.METRIC_L2, "IP": o.METRIC_IP, "cosine": o.METRIC_IP}[metric]
        xs, qs = (o.normalize_rows(x), o.normalize_rows(q)) if metric == "cosine" else (x, q)
        for suffix, al in (("", None), ("_f", alive)):
            oi, od = o.knn(qs, xs, k, om, alive=al)
            if metric == "cosine":
                od = (np.float32(1) - od).astype(np.float32)
            assert (rd("out_ids" + suffix, np.int64).reshape(nq, k) == oi).all()
            assert (rd("out_dis" + suffix, np.float32).reshape(nq, k).view(np.uint32) == od.view(np.uint32)).all()
        for tag, m in (("l2", o.METRIC_L2), ("ip", o.METRIC_IP)):
            oi, od = o.knn(q, x, k, m)
            assert (rd("bf_%s_ids" % tag, np.int64).reshape(nq, k) == oi).all()
            assert (rd("bf_%s_dis" % tag, np.float32).reshape(nq, k) == od).all()
        for tag, m in (("ham", o.METRIC_HAMMING), ("jac", o.METRIC_JACCARD)):
            oi, od = o.knn_bin(bq, bx, k, m)
            assert (rd("bf_%s_ids" % tag, np.int64).reshape(nb_q, k) == oi).all()
            assert (rd("bf_%s_dis" % tag, np.float32).reshape(nb_q, k) == od).all()
            # the BinaryFLAT index object (createVectorIndex<..., BinaryVector>: build from chunks, serialise, load, search)
            assert (rd("bi_%s_ids" % tag, np.int64).reshape(nb_q, k) == oi).all()
            assert (rd("bi_%s_dis" % tag, np.float32).reshape(nb_q, k) == od).all()
            fi, fd = o.knn_bin(bq, bx, k, m, alive=balive)
            assert (rd("bi_%s_ids_f" % tag, np.int64).reshape(nb_q, k) == fi).all()
            assert (rd("bi_%s_dis_f" % tag, np.float32).reshape(nb_q, k) == fd).all()


def test_text_shim_is_built_against_the_tantivy_stub():
    lib = os.path.join(ROOT, "shim", "_build", "libmsvs_text_shim.so")
    assert os.path.exists(lib) and os.path.exists(os.path.join(ROOT, "shim", "_build", "test_text_shim"))
    out = subprocess.check_output(["nm", "-D", "--defined-only", "-C", lib], text=True)
    for sym in ("TANTIVY::ffi_bm25_search", "TANTIVY::ffi_get_doc_freq", "TANTIVY::ffi_get_total_num_docs",
                "TANTIVY::ffi_get_total_num_tokens", "TANTIVY::ffi_load_index_reader",
                # the writer-side tee (TantivyIndexStore.cpp:713,742,792,824)
                "TANTIVY::ffi_create_index_with_parameter", "TANTIVY::ffi_index_multi_column_docs",
                "TANTIVY::ffi_index_writer_commit", "TANTIVY::ffi_free_index_writer"):
        assert sym in out, sym
    # the forwarding build (writer calls also go to the crate's bridge under another namespace) compiles
    subprocess.check_call(["g++", "-std=c++17", "-fsyntax-only", "-Istubs", "-DMSVS_TANTIVY_FORWARD_NS=TANTIVY_RS", "TextShim.cpp"],
                          cwd=os.path.join(ROOT, "shim"))


@pytest.mark.gpu
def test_text_shim_replays_the_two_part_bm25_golden():
    """Seam B through the functions the host really calls (TANTIVY::ffi_*, shim/TextShim.cpp): the documents of
    00041_mqvs_text_search_multiple_parts in two parts, each BUILT through ffi_create_index_with_parameter /
    ffi_index_multi_column_docs / ffi_index_writer_commit / ffi_free_index_writer, statistics summed over the parts like
    BM25InfoInDataParts, one ffi_bm25_search per part -> the golden hits; a filtered search; misuse is an error VALUE."""
    from golden_util import f32_of, load_goldens

    g = load_goldens()["00041_two_parts"]
    docs, n0 = g["docs"], g["part_sizes"][0]
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, "docs.txt"), "w") as f:
            for i, d_ in enumerate(docs):
                f.write("%d\t%s\n" % (0 if i < n0 else 1, " ".join(d_["texts"]).replace("\n", " ").replace("\t", " ")))
        with open(os.path.join(td, "query.txt"), "w") as f:
            f.write(g["text_query"] + "\nor\n")
        r = subprocess.run([os.path.join(ROOT, "shim", "_build", "test_text_shim"), td], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        lines = [ln.split() for ln in r.stdout.strip().split("\n")]
        assert os.path.exists(os.path.join(td, "part0", "postings.mspost"))
    hits = [(np.float32(ln[3]), docs[int(ln[2]) + (n0 if ln[1] == "1" else 0)]["id"]) for ln in lines if ln[0] == "hit"]
    hits.sort(key=lambda h: (-float(h[0]), h[1]))
    assert [h[1] for h in hits[:2]] == g["text_search_2parts"][0]
    assert np.array([h[0] for h in hits[:2]], np.float32).tolist() == f32_of(g["text_search_2parts"][1]).tolist()
    stats = [ln for ln in lines if ln[0] == "stats"][0]
    assert int(stats[1]) == len(docs)
    even = [ln for ln in lines if ln[0] == "even"]
    assert all(int(ln[2]) % 2 == 0 for ln in even) and {(ln[1], ln[2]) for ln in even} <= {(ln[1], ln[2]) for ln in lines if ln[0] == "hit"}
    assert ["missing_index_is_error", "1"] in lines
    assert ["freed_writer_is_error", "1"] in lines and ["other_tokenizer_is_error", "1"] in lines
    assert ["default_tokenizer_is_fine", "1"] in lines
    assert ["tokenizer_options_are_errors", "1"] in lines and ["default_options_spelled_out_are_fine", "1"] in lines
