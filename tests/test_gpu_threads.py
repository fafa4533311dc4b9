"""GPU: the library is re-entrant per database handle (SURVEY 8b "Threading").  Two host threads, each with its own
database handle, search their own batches at the same time, many times; every result equals the one a lone thread gets.
One handle used from two threads is serialised by the handle's lock and gives the same answers too."""
import threading

import pytest

pytestmark = pytest.mark.gpu


def test_two_threads_two_handles_and_one_shared_handle():
    import pangea_plus_amd as pg
    from pangea_plus_amd import _capi
    pg.init(0)
    cfgs = [pg.SynthCfg.default(n_seq=2500, seq_len=600, n_genus=50), pg.SynthCfg.default(n_seq=1800, seq_len=900, n_genus=30, seed=77)]
    dbs = [pg.Db.from_synth(c) for c in cfgs]
    dbs[1].set_ungapped(True)   # the two handles even run different specs
    batches = [[pg.Reads.from_synth(c, 1000 * k, 700 + 100 * k) for k in range(4)] for c in cfgs]
    want = [[_capi.blast_search(dbs[t], b).format(dbs[t], b) for b in batches[t]] for t in range(2)]
    assert all(len(w) > 50000 for ws in want for w in ws)
    errors = []

    def worker(t, db, rounds):
        try:
            for r in range(rounds):
                for k, b in enumerate(batches[t]):
                    got = _capi.blast_search(db, b).format(db, b)
                    if got != want[t][k]:
                        errors.append((t, r, k))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t, dbs[t], 6)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:5]
    # one handle, two threads: the handle's lock serialises the searches
    th = [threading.Thread(target=worker, args=(0, dbs[0], 4)) for _ in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:5]


def test_two_handles_search_one_batch_with_dust_inside_the_search():
    """with `pgx_db_set_dust_each_search` a search used to rewrite the DUST bits of the batch it was given, so
    two handles searching ONE batch from two threads raced (one zeroing the window bits the other's seed kernel was
    reading).  The in-search pass now writes into the handle's own workspace; the batch is read-only.  Reads with
    homopolymers and dinucleotide repeats, so that the bits matter; every answer equals the quiet single-thread one
    (which the low-complexity tests of test_gpu_blast.py pin against the checker)."""
    import random
    import pangea_plus_amd as pg
    from pangea_plus_amd import _capi
    pg.init(0)
    cfg = pg.SynthCfg.default(n_seq=1500, seq_len=700, n_genus=40)
    dbs = [pg.Db.from_synth(cfg) for _ in range(2)]
    rng = random.Random(5)
    base = pg.Reads.from_synth(cfg, 0, 1500)
    recs = []
    for i in range(1500):
        s = list("".join("ACGTN"[b] for b in base.get(i)))
        if i % 3 == 0:
            n = rng.choice([8, 12, 20, 40])
            p = rng.randrange(0, len(s) - n)
            s[p:p + n] = list((rng.choice(["A", "T", "AC", "GT", "CAG"]) * n)[:n])
        recs.append(">q%d\n%s\n" % (i, "".join(s)))
    batch = pg.Reads.from_fasta_text("".join(recs).encode())
    quiet = _capi.blast_search(dbs[0], batch).format(dbs[0], batch)
    dbs[0].set_dust(False)
    assert _capi.blast_search(dbs[0], batch).format(dbs[0], batch) != quiet  # the mask changes this batch's table
    dbs[0].set_dust(True)
    for d in dbs:
        d.set_dust_each_search(True)
    errors = []

    def worker(t):
        try:
            for r in range(12):
                if _capi.blast_search(dbs[t], batch).format(dbs[t], batch) != quiet:
                    errors.append((t, r))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))
    th = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors[:5]
