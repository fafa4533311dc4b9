// Internal object model of libpangea_hip: what lives in HBM for one process / one GPU.
#pragma once
#include <memory>
#include <functional>
#include <thread>
#include <atomic>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"
#include "rdp_host.hpp"

// ---------------------------------------------------------------------------------------------
// Sequence database in HBM.
//   d_words     2-bit packed bases of all sequences back to back (1 zero word in front, 2 behind)
//   d_amb       optional spaced ambiguity flags, same geometry (absent when the DB is pure ACGT)
//   d_seq_off   n_seq+1 base offsets;  d_blk_subj[b] = subject holding base 512*b
//   seed index  every 16-mer start position of the concatenation, grouped by
//               seed_bucket(kmer, index_bits):  d_bucket_off[2^bits + 1], d_postings[n_postings]
//               (ascending position inside a bucket).  At 1 Gbp: bits = 32 (17.2 GB of offsets +
//               4 GB of postings — sized for 288 GB of HBM, one dependent gather per probe).
// ---------------------------------------------------------------------------------------------
struct pgx_db {
	int64_t n_seq = 0, n_bases = 0;
	bool has_amb = false;
	bool synthetic_ids = false;
	std::vector<uint32_t> h_seq_off;
	std::vector<std::string> ids;
	// host copy of the packed bases is kept only for file-built databases (SOAP row formatting)
	std::vector<uint64_t> h_words, h_amb;

	pgx::DevBuf<uint64_t> d_words, d_amb;
	pgx::DevBuf<uint32_t> d_seq_off, d_blk_subj;
	pgx::DevBuf<uint4> d_blk_info; // per 512-base block: subject, its start, its end, the next subject's end
	pgx::DevBuf<uint32_t> d_amb_blk; // databases with ambiguity letters: one bit per 512-base block that holds one
	int index_bits = 0;
	int64_t n_postings = 0;
	std::shared_ptr<void> work; // classify.hip: the handle's search workspace (stream, tables, counters), made on first use
	std::mutex search_mu;       // searches through one handle are serialised; different handles share nothing
	bool dust = true;      // pgx_db_set_dust: `-dust no` switches the low-complexity mask of the reads off
	bool dust_each_search = false; // pgx_db_set_dust_each_search: S3d recomputed inside every search (as BLAST runs it), not only at import
	bool ungapped = false; // pgx_db_set_ungapped: searches through this handle stop after the ungapped stage (spec v1)
	pgx::DevBuf<uint32_t> d_bucket_off, d_postings;
	// databases without ambiguity: 12-byte records {posting, database bases left of the 16-mer, bases right of it}, the
	// stream k_seed_extend deals from (one contiguous piece per bucket instead of two)
	pgx::DevBuf<uint3> d_post_ctx;

	// taxonomy binding (pgx_db_bind_taxonomy): per subject lineage text + consensus tokens
	bool bound = false;
	std::vector<std::string> lineage;            // per subject, taxcollector text
	std::vector<std::string> token_text;         // token id -> text (id 0 = "")
	std::unordered_map<std::string, uint32_t> token_id;
	pgx::DevBuf<uint32_t> d_subj_tok_off, d_subj_tok; // CSR of token ids per subject
	pgx::DevBuf<int8_t> d_tok_rank;                   // token id -> index in "0".."6" or -1
	// 64-byte record per subject for the consensus kernel: [0] = ntok | npairs << 16 (npairs 0xFFFF: use the
	// CSR), [1..15] = (name id << 3 | rank index + 1) of each (rank,name) pair
	pgx::DevBuf<uint32_t> d_subj_pairs;
	int max_pairs = 15; // most pairs any subject record carries
	int pair_words = 16; // words per subject record: 8 (32 bytes) when no subject carries more than 7 pairs -- the usual seven-rank
			     // lineages --, else 16: the ordering kernel fetches one record per hit, and a record is a random line
	std::vector<int8_t> h_tok_rank;
	std::vector<int32_t> subj_taxid;
	// distinct lineage texts (the OTUs megaclust counts): id per subject, text per id; the last id is the empty text
	std::vector<std::string> lin_text;
	pgx::DevBuf<uint32_t> d_subj_lin;
	uint32_t empty_lin = 0;
	pgx::DevBuf<uint32_t> d_node_name_tok;       // taxid -> token id of a one-word scientific name (else 0)
	pgx::DevBuf<uint32_t> d_simrank_lut;         // pident hundredths -> string-order rank
	pgx::DevBuf<uint32_t> d_simrank_len;         // [length * 256 + mismatches] -> the same rank, alignments < 256 long
	uint32_t simrank_undef = 0, simrank_zero = 0;
	uint32_t intern(const std::string &s);
	// text tables of the formatters in HBM, made at the first table that needs them and kept with the handle (they used to be
	// put together and uploaded by every formatting call: 100 MB of lineage text per consensus file): subject ids, and the
	// distinct lineage texts of the bound taxonomy (made again after a new binding)
	mutable std::mutex fmt_mu;
	mutable pgx::DevBuf<unsigned char> d_id_blob, d_lin_blob;
	mutable pgx::DevBuf<uint32_t> d_id_off, d_lin_off;
	mutable bool id_blob_ready = false, lin_blob_ready = false;
};

// A batch of reads in HBM: forward and reverse-complement strands, each read word-aligned.
namespace pgx {
// Everything a DUST pass over a batch writes (dust.hip).  A batch owns one set (made at import); a database handle owns
// another for searches that compute S3d again (pgx_db_set_dust_each_search): a search never writes into the batch it is
// given, so two handles may search one batch from two threads.
struct DustBufs {
	DevBuf<uint64_t> win_f, win_r; // per strand one bit per read position: the 28 bases from there touch no masked base
				       // (written for the reads with a masked base only: nobody looks at the others' words)
	DevBuf<uint8_t> any;            // per read: it has a masked base
	DevBuf<uint64_t> mask;          // the masked bases of the listed reads
	DevBuf<uint32_t> list, n;       // reads the first pass listed, their number (on the device)
	DevBuf<uint2> range;            // first and last position of a listed read at which the algorithm's test passes
};
} // namespace pgx

struct pgx_reads {
	int64_t n = 0;
	int64_t first = 0; // ordinal of read 0 within its source (names r<first+i> for synthetic reads)
	bool synthetic = false;
	bool has_amb = false;
	// names of file-built batches: first word of each header, the batch's OWN compact copy: (offset, length) into h_text on
	// the host, d_name_at[i] .. d_name_at[i + 1] into d_names on the device (the formatters and the RDP import use those)
	std::shared_ptr<const pgx::TextBlob> h_text;
	std::vector<uint64_t> name_off;
	std::vector<uint32_t> name_len;
	pgx::DevBuf<unsigned char> d_names;
	pgx::DevBuf<uint32_t> d_name_at;
	std::vector<uint32_t> h_len, h_woff;
	std::vector<uint8_t> h_read_amb; // per read: holds an ambiguity letter (empty: none does)
	std::vector<uint8_t> h_read_dust; // per read: holds a DUST-masked base (empty: none does)
	// search classes (built once when the batch is made): reads are searched class by class -- flag words of a
	// diagonal (3: <= 192 bases, 5: <= 320, 8: <= 512, 0: longer) x ambiguity letters -- so that a few long or
	// N-holding reads do not slow down the rest.  One class without a list = the whole batch.
	struct SearchClass {
		int amb, words;
		uint32_t off, count;
		bool listed;
		int dust; // its reads have DUST-masked bases (S3d): the seed kernel variant that applies the window bits
	};
	std::vector<SearchClass> classes;
	pgx::DevBuf<uint32_t> d_class_list;
	std::vector<uint64_t> h_fwd; // host copy of the forward strand (file-built batches only)
	int64_t n_words = 0;
	int32_t max_len = 0;
	pgx::DevBuf<uint64_t> d_fwd, d_rc, d_fwd_amb, d_rc_amb;
	// spec v2 S3d (dust.hip): per strand one bit per read position: the 28 bases from there on touch no base that DUST
	// masks; 64 positions per word at the read's word offset; absent when no read of the batch has a masked base
	bool has_dust = false;
	pgx::DustBufs dustb; // what the batch's own DUST pass (at import; pgx_reads_redo_dust) wrote
	pgx::DevBuf<uint32_t> d_len, d_woff; // d_woff has n+1 entries
	// reads with a run of 6 or more unknown letters (mates joined by N's, Trim/trim2.4.pl:228-245) are searched as the
	// stretches between such runs (seqdb.hip: reads_build_pieces): `pieces` is a batch of its own, pieces of a read
	// are consecutive in it; null when no read of the batch has such a run
	std::unique_ptr<pgx_reads> pieces;
	pgx::DevBuf<uint32_t> d_piece_first;  // n+1: first piece of each read
	pgx::DevBuf<uint32_t> d_piece_parent; // per piece: its read
	pgx::DevBuf<uint32_t> d_piece_qoff;   // per piece: 0-based offset of its first base in the read
	std::string name_of(int64_t i) const;
};

struct pgx_hits {
	int64_t n_reads = 0;
	int64_t n_hits = 0; // slots in d_hits (including hits dropped by the 500-subject limit)
	bool gapped = true; // the search that made the table: spec v2 (gapped statistics, S4) or `-ungapped` (S4u) -- the formatter's
			    // e-value / bit-score columns follow the TABLE, not the handle's switch at format time
	pgx::DevBuf<pgx_hit> d_hits;     // grouped by read, spec order inside a read
	pgx::DevBuf<uint32_t> d_read_off; // n_reads+1 slot offsets
	pgx::DevBuf<uint32_t> d_read_cnt; // hits kept per read (<= slots of the read)
};

struct pgx_rdp {
	int64_t n = 0;
	pgx::DevBuf<uint32_t> d_off;   // n+1, in triplets
	pgx::DevBuf<uint32_t> d_name;  // token id of the cleaned name
	pgx::DevBuf<int8_t> d_rank;    // index in (domain..species) or -1
	pgx::DevBuf<uint32_t> d_code;  // name << 3 | rank + 1 (0xFFFFFFFF: matches nothing)
	pgx::DevBuf<uint8_t> d_present; // 0: read has no RDP line (never selected)
	int max_trip = 8;               // most triplets any read of the batch carries (bounds the agreement compare grid)
};

namespace pgx {

// (fnv64_bytes, NameIndexT, the host form of the RDP import: rdp_host.hpp, free of HIP)

// the names of a batch for NameIndexT
struct ReadsNames {
	const pgx_reads &rd;
	size_t size() const { return (size_t)rd.n; }
	void span(size_t i, std::string &tmp, const char **p, size_t *len) const
	{
		if (rd.synthetic) {
			tmp = rd.name_of((int64_t)i);
			*p = tmp.data();
			*len = tmp.size();
		} else {
			*p = rd.h_text->data() + rd.name_off[i];
			*len = rd.name_len[i];
		}
	}
};
struct ReadNameIndex : private ReadsNames, public NameIndexT<ReadsNames> {
	explicit ReadNameIndex(const pgx_reads &r) : ReadsNames{ r }, NameIndexT<ReadsNames>(static_cast<const ReadsNames &>(*this)) {}
};

// seqdb.hip
int db_upload_and_index(pgx_db *db);
int db_build_blk_info(pgx_db *db);
void index_check(const pgx_db *db, const char *where);
int reads_from_fasta_ex(const char *path, int64_t first, int64_t count, bool fold_to_g, std::vector<uint32_t> *amb_count,
			pgx_reads **out);
int db_fold_amb_to_g(const pgx_db *src, pgx_db **out);
int db_read_host(const char *prefix, pgx_db **out);
int64_t fasta_count_records(const char *path);
int64_t fasta_count_records_text(const char *base, size_t len, bool *at_line_start);
int reads_from_fasta_text(std::shared_ptr<const TextBlob> text, int64_t first, int64_t count, bool fold_to_g,
			  std::vector<uint32_t> *amb_count, pgx_reads **out);
int db_build_index(pgx_db *db);
int choose_index_bits(int64_t n_postings);

// trim.hip
int device_line_index(const uint8_t *d_text, uint64_t n, bool open_tail, DevBuf<uint64_t> &start, uint64_t *n_lines_out);

// annotate.hip / rdp_device.hip
// the RDP table of a batch parsed on the device; 1 = this form does not apply to the batch (the caller takes the host form)
int rdp_from_text_device(const char *text, size_t n_bytes, const pgx_reads *reads, pgx_db *db, pgx_rdp **out);

// dust.hip
int reads_dust(pgx_reads *rd);
int reads_dust_again(const pgx_reads *rd, DustBufs &b, hipStream_t stream);

// classify.hip
struct SearchCounters {
	unsigned long long probes, postings, candidates, hits;
};
int blast_search_device(pgx_db *db, pgx_reads *reads, pgx_hits *out, pgx_stage_times *times);
int consensus_device(const pgx_db *db, const pgx_hits *hits, const pgx_rdp *rdp, pgx_consensus_rec *d_out,
		     pgx_stage_times *times);

// bigreads.hip: spec order + 500-subject cut for reads with more than 64 hits
int sort_big_reads(pgx_hit *hits, const pgx_hit *scratch, const uint32_t *read_start, const uint32_t *off, uint32_t *read_cnt,
		   const uint32_t *big_list, uint32_t n_big, bool gapped);

// columns of a hit that follow from the stored ones (include/pangea_hip.h: pgx_hit)
__host__ __device__ inline int hit_qspan(const pgx_hit &h) { return h.qend - h.qstart + 1; }
__host__ __device__ inline int hit_sspan(const pgx_hit &h) { return (h.send > h.sstart ? h.send - h.sstart : h.sstart - h.send) + 1; }
__host__ __device__ inline int hit_diffs(const pgx_hit &h)
{
	const int t = hit_qspan(h) + hit_sspan(h);
	return (t - 2 * h.score - (t & 1)) / 6;
}
__host__ __device__ inline int hit_gaps(const pgx_hit &h) { return hit_diffs(h) - (int)h.mismatch; }
__host__ __device__ inline int hit_length(const pgx_hit &h) { return (hit_qspan(h) + hit_sspan(h) + hit_gaps(h)) / 2; }

// Stage probes (PGX_SEED_STOP / PGX_SORT_STOP truncate the kernels after a stage: tools/probe_stages.py) exist only in
// builds made with -DPGX_STAGE_PROBES; the shipped kernels carry no such branches.
#ifdef PGX_STAGE_PROBES
#define PGX_DBG_STOP(v) ((v).dbg_stop)
#define PGX_SORT_DBG(v) ((v).dbg)
#else
#define PGX_DBG_STOP(v) 0
#define PGX_SORT_DBG(v) 0
#endif

struct DbView {
	const uint64_t *words, *amb;
	const uint32_t *seq_off, *blk_subj, *bucket_off, *postings;
	const uint4 *blk_info;
	const uint3 *post_ctx;
	const uint32_t *amb_blk; // one bit per 512-base block with an ambiguity letter (null: the database has none)
	uint32_t n_seq;
	int64_t n_bases; // letters of all subjects (the packed words hold 768 zero letters in front and behind)
	int bits;
	int gapped;   // spec v2: the seed stage hands over initial HSPs (score field = offset of the seed run in the HSP) to gapped.hip
	int deep_from; // reads longer than this run the gapped stage's 40-difference rows (their ordering key counts mismatches)
	int dbg_stop; // profiling aid (PGX_SEED_STOP): 1 = probes only, 2 = + postings/filter, 3 = + queue without diagonal work
};

struct ReadsView {
	const uint64_t *fwd, *rc, *fwd_amb, *rc_amb;
	const uint64_t *dustwin_f, *dustwin_r; // S3d window bits per strand (null: no read of the batch is masked, or `-dust no`)
	const uint8_t *dust_any;               // per read: has a masked base
	const uint32_t *len, *woff;
	uint32_t n;           // reads this launch works on
	const uint32_t *list; // their ids (null: 0 .. n-1): a batch is searched class by class (flag words, ambiguity)
};


constexpr uint32_t kFragmented = 0xFFFFFFFFu; // read_start of a read whose hits went to the overflow table

// gapped.hip: spec v2 S3b on the initial HSPs of a seed-stage table (main table addressed through read_start / read_cnt,
// overflow table flat), in place; long reads and extensions with many differences go through `big` lists
struct GappedWork {
	DevBuf<unsigned long long> big_list, big_list2; // HSPs for the one-wavefront-per-HSP kernel; those of them that need its large rows
	DevBuf<uint32_t> big_count; // [0] entries appended (may exceed the capacity: the caller grows and repeats), [1] of the second list
	DevBuf<uint2> side_main, side_ovf, side_list; // per table slot (per list entry, second tier): the left side's extension, parked until the right side is done
	DevBuf<uint32_t> order;            // per block of k_gapped_fast: the pool's HSPs in cost order
	DevBuf<uint32_t> items, bins;      // binned form: the main table's slots in (region, bin) order; histograms, cursors, total
	DevBuf<uint32_t> items1;           // ... in region order (first pass), with their key bytes
	DevBuf<uint8_t> keys1;
};
int gapped_stage(const DbView &dv, const ReadsView &rv, pgx_hit *main_table, const uint8_t *main_key, const uint32_t *read_start,
		 const uint32_t *read_cnt, pgx_hit *ovf_table, const uint8_t *ovf_key, const unsigned long long *ovf_count, unsigned long long ovf_cap, bool long_reads,
		 unsigned long long hit_cap, int max_len, GappedWork &gw, hipStream_t stream, const unsigned long long *main_used, const uint8_t *main_reg);
// reads longer than this many bases run the gapped stage's rows for 40 differences a side (PGX_GAP_DEEP_FROM: measurement aid)
inline int gapped_deep_from()
{
	static const int v = [] {
		const char *e = getenv("PGX_GAP_DEEP_FROM");
		const int x = e ? atoi(e) : 320;
		return x < 192 ? 192 : (x > 320 ? 320 : x);
	}();
	return v;
}
// the main table's HSPs are handled region by region of the database (at most 240 regions: positions >> this)
inline int gapped_region_shift(int64_t n_bases)
{
	int s = 0;
	while (((n_bases + 1024) >> s) > 240)
		s++;
	return s;
}

// pident as printf("%.2f", 100.0*m/L) would print it, in hundredths (exact, ties via the double)
__host__ __device__ inline int pident_hundredths(int matches, int length)
{
	long long num = 10000ll * matches;
	long long q = num / length, r = num % length;
	if (2 * r > length)
		return (int)(q + 1);
	if (2 * r < length)
		return (int)q;
	// exact rational tie: printf rounds the binary value d = 100.0*m/L, which may sit on either
	// side of it; fma gives the sign of d*200 - (2q+1) exactly
	double d = 100.0 * (double)matches / (double)length;
	double s = fma(d, 200.0, -(double)(2 * q + 1));
	if (s > 0)
		return (int)(q + 1);
	if (s < 0)
		return (int)q;
	return (int)(q + (q & 1));
}

} // namespace pgx
