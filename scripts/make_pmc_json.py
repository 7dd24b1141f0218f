#!/usr/bin/env python3
"""Build profiles/rNN_pmc_schur.json (bench.py reads profiles/r05_pmc_schur.json) from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; rocpd SQLite) of
`python bench.py --steps 1 --warmup 2 --no-cpu-baseline` (4 factorisations per run: counted from the k_scatter_values launches).
usage: make_pmc_json.py fetch.db write.db n_factorisations "source text" > profiles/r01_pmc_schur.json"""
import hashlib, json, os, re, sqlite3, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """the same hash bench.py checks: the traffic figure is only valid for the kernels it was measured on"""
    h = hashlib.sha256()
    for f in ("sluamd_kernels.hip", "sluamd_zkernels.inc"):
        h.update(open(os.path.join(ROOT, "superlu_dist_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def total(path, counter):
    cur = sqlite3.connect(path).cursor()
    s = 0.0; n = 0
    for name, cname, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
        if cname == counter and "k_schur" in name:
            s += val; n += 1
    return s, n


def count_launches(path, counter, what):
    cur = sqlite3.connect(path).cursor()
    return sum(1 for name, cname in cur.execute("select name, counter_name from pmc_events") if cname == counter and what in name)


fetch_kb, nl = total(sys.argv[1], "FETCH_SIZE")
write_kb, nl2 = total(sys.argv[2], "WRITE_SIZE")
# factorisations in the run: every bench step (warm-up, timed, profiled) re-distributes A with one k_scatter_values launch, and the
# handle creation launches one more; argv[3] is only the fallback
nsc = count_launches(sys.argv[1], "FETCH_SIZE", "k_scatter_values")
nf = nsc - 1 if nsc > 1 else int(sys.argv[3])
fetch_b, write_b = fetch_kb * 1024.0, write_kb * 1024.0
out = {
    "source": sys.argv[4],
    "kernel_source_sha16": kernel_source_hash(),
    "kernel": "k_schur<128,128,8> + k_schur<64,64,4> (all launches of a factorisation)",
    "factorisations_in_run": nf, "launches_in_run": nl,
    "fetch_bytes_per_factorisation_raw": fetch_b / nf, "write_bytes_per_factorisation_raw": write_b / nf,
    "note": "FETCH_SIZE/WRITE_SIZE are in KB. MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide "
            "coalesced streaming read -> fetch x2 applied in traffic_bytes_per_factorisation; WRITE_SIZE used as is.",
    "traffic_bytes_per_factorisation": (2.0 * fetch_b + write_b) / nf,
}
print(json.dumps(out, indent=1))
