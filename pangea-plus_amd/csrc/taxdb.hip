// Taxonomy database: the reference's Tax_class/ncbitc.c as a library.
//
//   pgx_tax_create         text dumps -> gi_taxid_nucl.dmp.bin / nodes.dmp.bin / names.dmp.bin,
//                          byte-compatible with ncbitc.c:701-839 (struct layouts ncbitc.c:98-140)
//   pgx_tax_open           loads the three .bin files and puts gi->taxid, parent, tax_id and the
//                          driver-rank code of every node in HBM
//   single lookups         ncbitc.c:567-699 (host: O(1) table reads; no kernel is worth launching)
//   pgx_tax_lineage_batch  the per-hit walk the Perl driver performs with ~15 fork/exec of
//                          ./tax_class (NCBI-taxcollector-0.01.pl:166-300) as one HIP kernel
#include <getopt.h>

#include <algorithm>

#include "taxdb.hpp"

namespace pgx {

// enum order of ncbitc.c:39-69
static const char *const kRankNames[] = { "class",      "family",        "forma",          "genus",      "infraclass",
					  "infraorder", "kingdom",       "no rank",        "order",      "parvorder",
					  "phylum",     "species",       "species group",  "species subgroup", "subclass",
					  "subfamily",  "subgenus",      "subkingdom",     "suborder",   "subphylum",
					  "subspecies", "subtribe",      "superclass",     "superfamily", "superkingdom",
					  "superorder", "superphylum",   "tribe",          "varietas" };
static const int kNumRanks = (int)(sizeof(kRankNames) / sizeof(kRankNames[0]));

const char *tax_rank_text(int id)
{
	return (id >= 0 && id < kNumRanks) ? kRankNames[id] : "invalid id";
}

static int tax_rank_code(const char *s)
{
	for (int i = 0; i < kNumRanks; i++)
		if (strncmp(s, kRankNames[i], 32) == 0)
			return i;
	return -1;
}

// rank enum -> what the Perl driver does with it (NCBI-taxcollector-0.01.pl:228-237, 261, 285):
// 0..7 = index in (superkingdom phylum class order family genus species kingdom),
// 8 = "no rank", 9 = any other rank (walk on)
int8_t driver_rank_code(int rank_enum)
{
	switch (rank_enum) {
	case 24: return 0; // superkingdom
	case 10: return 1; // phylum
	case 0: return 2;  // class
	case 8: return 3;  // order
	case 1: return 4;  // family
	case 3: return 5;  // genus
	case 11: return 6; // species
	case 6: return 7;  // kingdom
	case 7: return 8;  // no rank
	default: return 9;
	}
}

static void wr32(uint8_t *p, int32_t v)
{
	memcpy(p, &v, 4); // x86-64 / gfx950 hosts are little endian, as the reference assumes
}

static int32_t rd32(const uint8_t *p)
{
	int32_t v;
	memcpy(&v, p, 4);
	return v;
}

// the reference trims a scanned field in place: first char -> ' ', last char dropped, and
// anything of length <= 2 becomes empty (ncbitc.c:495-509)
static void trim_field(char *s, int max)
{
	int len = (int)strnlen(s, (size_t)max);
	if (len <= 2) {
		s[0] = '\0';
	} else {
		s[0] = ' ';
		s[len - 1] = '\0';
	}
}

// next '|'-terminated field of a dump line starting at *p; leading blanks of NUMERIC fields are
// skipped by the caller. Returns false when no terminator is left.
static bool take_field(const char *&p, std::string &out)
{
	const char *bar = strchr(p, '|');
	if (!bar)
		return false;
	out.assign(p, (size_t)(bar - p));
	p = bar + 1;
	return true;
}

static bool parse_int_field(const std::string &f, int &v)
{
	// sscanf("%d |") semantics: optional blanks, a decimal integer, optional blanks
	const char *s = f.c_str();
	char *end;
	long x = strtol(s, &end, 10);
	if (end == s)
		return false;
	while (*end == ' ' || *end == '\t' || *end == '\n')
		end++;
	if (*end)
		return false;
	v = (int)x;
	return true;
}

// nodes.dmp line -> 28-byte record. The reference scans every %d through an int* into char/short
// members in ascending address order (ncbitc.c:517-525): of each 4-byte store only the bytes that
// the next store does not overwrite survive. Fields that fail to scan keep the previous line's
// bytes, as the reference's reused stack struct does.
static void node_record_from_line(const char *line, uint8_t rec[28])
{
	const char *p = line;
	std::string f[13];
	int nf = 0;
	while (nf < 13 && take_field(p, f[nf]))
		nf++;
	int v;
	if (nf > 0 && parse_int_field(f[0], v))
		wr32(rec, v);
	else
		return;
	if (nf > 1 && parse_int_field(f[1], v))
		wr32(rec + 4, v);
	else
		return;
	if (nf < 4 || f[2].empty() || f[3].empty())
		return; // %[^|] needs at least one character
	bool numeric_ok = true;
	const int at[8] = { 12, 14, 16, 18, 20, 24, 25, 26 };
	const int width[8] = { 2, 2, 2, 2, 4, 1, 1, 2 };
	for (int k = 0; k < 8 && numeric_ok; k++) {
		if (nf > 4 + k && parse_int_field(f[4 + k], v)) {
			for (int b = 0; b < width[k]; b++)
				rec[at[k] + b] = (uint8_t)((uint32_t)v >> (8 * b));
		} else {
			numeric_ok = false;
		}
	}
	char rank[300], embl[300];
	snprintf(rank, sizeof rank, "%s", f[2].c_str());
	snprintf(embl, sizeof embl, "%s", f[3].c_str());
	trim_field(rank, 32);
	wr32(rec + 8, tax_rank_code(rank + 1)); // int store over bytes 8..11 (ncbitc.c:528)
	trim_field(embl, 32);
	if (embl[0] != '\0') {
		rec[9] = (uint8_t)embl[1];
		rec[10] = (uint8_t)embl[2];
		rec[11] = 0;
	} else {
		rec[9] = 0;
	}
}

// names.dmp line -> 196-byte record (ncbitc.c:547-557). An over-long field runs into the next
// member, which its own scan then overwrites: the net effect is truncation at 63 characters.
static void name_record_from_line(const char *line, uint8_t rec[196])
{
	const char *p = line;
	std::string f[4];
	int nf = 0;
	while (nf < 4 && take_field(p, f[nf]))
		nf++;
	int v;
	if (nf < 1 || !parse_int_field(f[0], v))
		return;
	wr32(rec, v);
	for (int k = 0; k < 3; k++) {
		if (nf < 2 + k || f[1 + k].empty())
			return;
		const size_t room = (size_t)(196 - 4 - 64 * k);
		const size_t n = std::min(f[1 + k].size() + 1, room);
		memcpy(rec + 4 + 64 * k, f[1 + k].c_str(), n);
	}
	trim_field((char *)rec + 4, 64);
	trim_field((char *)rec + 68, 64);
	trim_field((char *)rec + 132, 32);
}

static std::string join_path(const char *dir, const char *name)
{
	std::string p = dir && *dir ? dir : ".";
	if (p.back() != '/')
		p += '/';
	return p + name;
}

struct LineReader {
	// fgets(line, 512) semantics of the reference (NCBITC_LINE_SIZE, ncbitc.c:74): longer lines
	// are cut into 511-byte pieces
	FILE *f;
	char buf[512];
	explicit LineReader(FILE *fp) : f(fp) {}
	bool next() { return fgets(buf, sizeof buf, f) != nullptr; }
};

// ------------------------------------------------------------------------------------------ device walk
// status: 0 ok, 1 gi has taxid 0 ("Unidentified(GI:n);"), 2 reference never terminates on this gi
// (leaf is the root or a child of the root, a failed node lookup, or a parent cycle), 3 more than
// PGX_LINEAGE_SLOTS elements
__global__ void k_tax_walk(const int32_t *__restrict__ gi, int64_t n, const int32_t *__restrict__ gi2tax, int64_t n_gi,
			   const int32_t *__restrict__ node_taxid, const int32_t *__restrict__ node_parent,
			   const int8_t *__restrict__ node_code, int64_t n_nodes, int32_t *__restrict__ lineage,
			   int32_t *__restrict__ count, int32_t *__restrict__ status, int32_t *__restrict__ leaf_out)
{
	int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n)
		return;
	const int32_t g = gi[i];
	int32_t *lin = lineage + i * PGX_LINEAGE_SLOTS;
	int cnt = 0, st = 0;
	int32_t leaf = 0;
	if (g <= 0) {
		st = 2; // negative file offset: "Error." (ncbitc.c:581-585)
	} else {
		int32_t t = (int64_t)g - 1 < n_gi ? gi2tax[g - 1] : 0;
		if (t == 0) {
			st = 1;
		} else {
			// `tax_class -s` prints nothing for the root or a child of the root (ncbitc.c:943-953)
			int32_t lt = 0, lp = 0;
			if (t > 0 && (int64_t)t - 1 < n_nodes) {
				lt = node_taxid[t - 1];
				lp = node_parent[t - 1];
			}
			if (t < 0 || t == 1 || lp == 1) {
				st = 2;
			} else {
				leaf = lt;
				t = lt;
				for (int depth = 0;; depth++) {
					if (t <= 0 || depth > 4096) {
						st = 2;
						break;
					}
					int32_t par = 0;
					int8_t code = 2; // a record past the end reads as zeros: rank enum 0 = "class"
					if ((int64_t)t - 1 < n_nodes) {
						par = node_parent[t - 1];
						code = node_code[t - 1];
					}
					if (code < 8) {
						if (cnt >= PGX_LINEAGE_SLOTS) {
							st = 3;
							break;
						}
						lin[cnt++] = t;
						if (code == 0)
							break; // superkingdom reached
					} else if (code == 8 && par == 1) {
						if (cnt >= PGX_LINEAGE_SLOTS) {
							st = 3;
							break;
						}
						lin[cnt++] = PGX_LIN_UNCLASSIFIED;
						break;
					}
					t = par;
				}
			}
		}
	}
	count[i] = st == 2 || st == 3 ? 0 : cnt;
	status[i] = st;
	if (leaf_out)
		leaf_out[i] = leaf;
}

int tax_walk_device(pgx_taxdb *db, const int32_t *d_gi, int64_t n, int32_t *d_lineage, int32_t *d_count, int32_t *d_status,
		    int32_t *d_leaf)
{
	if (n == 0)
		return 0;
	hipLaunchKernelGGL(k_tax_walk, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, d_gi, n, db->d_gi2tax.data(),
			   (int64_t)db->gi2tax.size(), db->d_node_taxid.data(), db->d_node_parent.data(),
			   db->d_node_code.data(), (int64_t)db->n_nodes, d_lineage, d_count, d_status, d_leaf);
	PGX_HIP(hipGetLastError());
	return 0;
}

// ------------------------------------------------------------------------------------------ host lookups
int tax_node_record(const pgx_taxdb *db, int taxid, pgx_node *out)
{
	memset(out, 0, sizeof *out);
	if (!db->have_nodes)
		return -1;
	if ((long)taxid - 1 < 0)
		return -2;
	if ((size_t)taxid - 1 >= db->n_nodes)
		return 0; // past EOF: fread fails, the record reads as zeros (observed)
	memcpy(out, db->nodes.data() + (size_t)(taxid - 1) * 28, 28);
	return 0;
}

// names lookup with the reference's exact probing order (ncbitc.c:647-699): the bisection runs
// over positions 0..num-1 while records are numbered 1..num, so the last record is never found; and a
// search that probes position 0 (a negative file offset: ncbitc_seek_name closes the file, :637-640, and every
// later seek fails on the closed stream) finds nothing from then on — whether `-n 1` gets there depends on the
// number of records.  `failed_seeks` = the perror lines the reference prints in such a search.
int tax_names_lookup(const pgx_taxdb *db, int taxid, std::vector<const uint8_t *> &out, int *failed_seeks)
{
	if (failed_seeks)
		*failed_seeks = 0;
	if (!db->have_names)
		return -1;
	const int num = db->n_names;
	const uint8_t *cur = nullptr;
	int closed = 0;
	auto seek = [&](int pos) -> int {
		if (closed || pos <= 0) {
			closed++;
			return -1;
		}
		if ((size_t)(pos - 1) < db->names_records)
			cur = db->names.data() + 4 + (size_t)(pos - 1) * 196;
		return cur ? rd32(cur) : 0;
	};
	int lo = 0, hi = num - 1, j = 0;
	bool hit = false;
	while (lo <= hi) {
		j = (lo + hi) / 2;
		int v = seek(j);
		if (v == taxid) {
			hit = true;
			break;
		}
		if (v < taxid)
			lo = j + 1;
		else
			hi = j - 1;
	}
	if (failed_seeks)
		*failed_seeks = closed;
	if (!hit)
		return 0;
	int i;
	for (i = j - 1; i > 0; i--)
		if (seek(i) != taxid)
			break;
	for (i = i + 1; i < num; i++) {
		if (seek(i) != taxid)
			break;
		out.push_back(cur);
	}
	return 1;
}

static void format_node_text(const pgx_node *n, Text &out)
{
	char embl[4] = { n->embl_code[0], n->embl_code[1], n->embl_code[2], 0 };
	out.printf("%d | %d | %s | %s | %d | %d | %d | %d | %d | %d | %d | %d | %s |\n", n->tax_id, n->parent_tax_id,
		   tax_rank_text(n->rank), embl, n->division_id, n->inherited_div_flag, n->genetic_code_id,
		   n->inherited_GC_flag, n->mitochondrial_genetic_code_id, n->inherited_MGC_flag, n->GenBank_hidden_flag,
		   n->hidden_subtree_root_flag, "");
}

static void format_name_text(const uint8_t *rec, Text &out)
{
	out.printf("%d | %s | %s | %s |\n", rd32(rec), (const char *)rec + 4, (const char *)rec + 68,
		   (const char *)rec + 132);
}

// first "scientific name" of a taxid as the driver extracts it (NCBI-taxcollector-0.01.pl:188-224):
// returns false when `tax_class -n` shows none (which includes the never-found last record)
bool tax_scientific_name(const pgx_taxdb *db, int taxid, std::string &name)
{
	std::vector<const uint8_t *> recs;
	if (tax_names_lookup(db, taxid, recs) <= 0)
		return false;
	for (const uint8_t *r : recs) {
		const char *cls = (const char *)r + 132;
		// the class field is matched as a substring of the printed " <class> " column
		if (!strstr(cls, "scientific name"))
			continue;
		std::string s((const char *)r + 4);
		s.erase(std::remove(s.begin(), s.end(), '\t'), s.end());
		auto ws = [](char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; };
		size_t b = 0, e = s.size();
		while (b < e && ws(s[b]))
			b++;
		while (e > b && ws(s[e - 1]))
			e--;
		name = s.substr(b, e - b);
		return true;
	}
	return false;
}

} // namespace pgx

using namespace pgx;

extern "C" {

int pgx_tax_create(const char *dir)
{
	// gi_taxid_nucl.dmp -> dense int32 table indexed by gi-1, holes zero-filled (ncbitc.c:701-748)
	{
		FILE *fi = fopen(join_path(dir, "gi_taxid_nucl.dmp").c_str(), "r");
		if (fi) {
			FILE *fo = fopen(join_path(dir, "gi_taxid_nucl.dmp.bin").c_str(), "wb");
			if (!fo) {
				fclose(fi);
				return fail(PGX_E_IO, "cannot write gi_taxid_nucl.dmp.bin in %s", dir);
			}
			LineReader lr(fi);
			int last = 0, gi = 0, tax = 0;
			const int32_t zero = 0;
			while (lr.next()) {
				sscanf(lr.buf, "%d\t%d", &gi, &tax);
				for (int k = 1; k < gi - last; k++)
					fwrite(&zero, 4, 1, fo);
				int32_t t = tax;
				fwrite(&t, 4, 1, fo);
				last = gi;
			}
			fclose(fo);
			fclose(fi);
		}
	}
	// nodes.dmp -> 28-byte records at index taxid-1 (ncbitc.c:750-794)
	{
		FILE *fi = fopen(join_path(dir, "nodes.dmp").c_str(), "r");
		if (fi) {
			FILE *fo = fopen(join_path(dir, "nodes.dmp.bin").c_str(), "wb");
			if (!fo) {
				fclose(fi);
				return fail(PGX_E_IO, "cannot write nodes.dmp.bin in %s", dir);
			}
			LineReader lr(fi);
			uint8_t rec[28] = { 0 }, zero[28] = { 0 };
			int last = 0;
			while (lr.next()) {
				node_record_from_line(lr.buf, rec);
				int tax = rd32(rec);
				for (int k = 1; k < tax - last; k++)
					fwrite(zero, 28, 1, fo);
				fwrite(rec, 28, 1, fo);
				last = tax;
			}
			fclose(fo);
			fclose(fi);
		}
	}
	// names.dmp -> record count + 196-byte records in file order (ncbitc.c:796-839)
	{
		FILE *fi = fopen(join_path(dir, "names.dmp").c_str(), "r");
		if (fi) {
			FILE *fo = fopen(join_path(dir, "names.dmp.bin").c_str(), "wb");
			if (!fo) {
				fclose(fi);
				return fail(PGX_E_IO, "cannot write names.dmp.bin in %s", dir);
			}
			LineReader lr(fi);
			uint8_t rec[196] = { 0 };
			int32_t num = 0;
			fwrite(&num, 4, 1, fo);
			while (lr.next()) {
				name_record_from_line(lr.buf, rec);
				fwrite(rec, 196, 1, fo);
				num++;
			}
			rewind(fo);
			fwrite(&num, 4, 1, fo);
			fclose(fo);
			fclose(fi);
		}
	}
	return 0;
}

static bool slurp(const std::string &path, std::vector<uint8_t> &out)
{
	FILE *f = fopen(path.c_str(), "rb");
	if (!f)
		return false;
	fseek(f, 0, SEEK_END);
	long sz = ftell(f);
	fseek(f, 0, SEEK_SET);
	out.resize(sz > 0 ? (size_t)sz : 0);
	size_t got = out.empty() ? 0 : fread(out.data(), 1, out.size(), f);
	fclose(f);
	out.resize(got);
	return true;
}

// host part of open: usable without a device for the single lookups of the CLI
static pgx_taxdb *tax_load(const char *dir)
{
	pgx_taxdb *db = new pgx_taxdb();
	std::vector<uint8_t> raw;
	if (slurp(join_path(dir, "gi_taxid_nucl.dmp.bin"), raw)) {
		db->have_gi = true;
		db->gi2tax.resize(raw.size() / 4);
		memcpy(db->gi2tax.data(), raw.data(), db->gi2tax.size() * 4);
	}
	if (slurp(join_path(dir, "nodes.dmp.bin"), db->nodes)) {
		db->have_nodes = true;
		db->n_nodes = db->nodes.size() / 28;
	}
	if (slurp(join_path(dir, "names.dmp.bin"), db->names) && db->names.size() >= 4) {
		db->have_names = true;
		db->n_names = rd32(db->names.data());
		db->names_records = (db->names.size() - 4) / 196;
	}
	return db;
}

int pgx_tax_open(const char *dir, pgx_taxdb **out)
{
	if (!dir || !out)
		return fail(PGX_E_ARG, "pgx_tax_open: null argument");
	PGX_TRY(require_device());
	pgx_taxdb *db = tax_load(dir);
	if (!db->have_gi || !db->have_nodes || !db->have_names) {
		delete db;
		return fail(PGX_E_IO, "taxonomy binaries missing in %s (run tax_class -c)", dir);
	}
	std::vector<int32_t> tid(db->n_nodes), par(db->n_nodes);
	std::vector<int8_t> code(db->n_nodes);
	for (size_t i = 0; i < db->n_nodes; i++) {
		const uint8_t *r = db->nodes.data() + i * 28;
		tid[i] = rd32(r);
		par[i] = rd32(r + 4);
		code[i] = driver_rank_code((int8_t)r[8]);
	}
	int rc = db->d_gi2tax.alloc(db->gi2tax.size());
	if (rc == 0)
		rc = db->d_gi2tax.upload(db->gi2tax.data(), db->gi2tax.size());
	if (rc == 0)
		rc = db->d_node_taxid.alloc(db->n_nodes);
	if (rc == 0)
		rc = db->d_node_taxid.upload(tid.data(), tid.size());
	if (rc == 0)
		rc = db->d_node_parent.alloc(db->n_nodes);
	if (rc == 0)
		rc = db->d_node_parent.upload(par.data(), par.size());
	if (rc == 0)
		rc = db->d_node_code.alloc(db->n_nodes);
	if (rc == 0)
		rc = db->d_node_code.upload(code.data(), code.size());
	if (rc < 0) {
		delete db;
		return rc;
	}
	*out = db;
	return 0;
}

void pgx_tax_close(pgx_taxdb *db) { delete db; }

int pgx_tax_gi2taxid(const pgx_taxdb *db, int gi, int *taxid)
{
	if (!db || !taxid)
		return fail(PGX_E_ARG, "pgx_tax_gi2taxid: null argument");
	if (!db->have_gi)
		return fail(PGX_E_IO, "fopen: No such file or directory");
	if ((long)gi - 1 < 0)
		return fail(PGX_E_ARG, "fseek: Invalid argument");
	*taxid = (size_t)gi - 1 < db->gi2tax.size() ? db->gi2tax[(size_t)gi - 1] : 0;
	return 0;
}

int pgx_tax_node(const pgx_taxdb *db, int taxid, pgx_node *out)
{
	if (!db || !out)
		return fail(PGX_E_ARG, "pgx_tax_node: null argument");
	int rc = tax_node_record(db, taxid, out);
	if (rc == -1)
		return fail(PGX_E_IO, "fopen: No such file or directory");
	if (rc == -2)
		return fail(PGX_E_ARG, "fseek: Invalid argument");
	return 0;
}

int pgx_tax_names(const pgx_taxdb *db, int taxid, pgx_name *buf, int cap)
{
	if (!db)
		return fail(PGX_E_ARG, "pgx_tax_names: null argument");
	std::vector<const uint8_t *> recs;
	int rc = tax_names_lookup(db, taxid, recs);
	if (rc < 0)
		return fail(PGX_E_IO, "fopen: No such file or directory");
	for (size_t i = 0; i < recs.size() && (int)i < cap && buf; i++)
		memcpy(&buf[i], recs[i], 196);
	return (int)recs.size();
}

int pgx_tax_format_node(const pgx_node *n, char *buf, size_t cap)
{
	if (!n || !buf)
		return fail(PGX_E_ARG, "pgx_tax_format_node: null argument");
	Text t;
	format_node_text(n, t);
	snprintf(buf, cap, "%s", t.s.c_str());
	return (int)t.s.size();
}

int pgx_tax_format_name(const pgx_name *n, char *buf, size_t cap)
{
	if (!n || !buf)
		return fail(PGX_E_ARG, "pgx_tax_format_name: null argument");
	Text t;
	format_name_text((const uint8_t *)n, t);
	snprintf(buf, cap, "%s", t.s.c_str());
	return (int)t.s.size();
}

// ncbitc.c:860-1004 — option scan, last verb wins, help on anything unknown, "Error." + exit 255
// when a lookup fails. Single lookups are table reads on the host; nothing here needs the GPU.
int pgx_tax_cli(int argc, char **argv, const char *dir, char **out_text, char **err_text)
{
	static const struct option lopts[] = { { "help", no_argument, 0, 'h' },
					       { "verbose", no_argument, 0, 'v' },
					       { "create", no_argument, 0, 'c' },
					       { "search", required_argument, 0, 's' },
					       { "search-gi", required_argument, 0, 'g' },
					       { "search-name", required_argument, 0, 'n' },
					       { "search-node", required_argument, 0, 't' },
					       { 0, 0, 0, 0 } };
	static const char *const kHelp = "Usage: ncbitc [options]\n"
					 "Options:\n"
					 "   -s --search id         search all tree using a gi index\n"
					 "   -g --search-gi id      search a tax id using a gi index\n"
					 "   -t --search-node id    search a node entry using a tax id\n"
					 "   -n --search-name id    search a name entry using a tax id\n"
					 "   -v --verbose           turn on verbose output\n"
					 "   -h --help              print this help message\n";
	Text out, err;
	int status = 0;
	char verb = 0;
	int index = 0, tax_id = 0, verbose = 0, c;
	bool helped = false;
	optind = 0;
	opterr = 0;
	while (!helped && (c = getopt_long(argc, argv, "hcs:n:vt:g:", lopts, nullptr)) != -1) {
		switch (c) {
		case 'c': verb = 'c'; break;
		case 's': verb = 's'; index = atoi(optarg); break;
		case 'g': verb = 'g'; index = atoi(optarg); break;
		case 'n': verb = 'n'; tax_id = atoi(optarg); break;
		case 't': verb = 't'; tax_id = atoi(optarg); break;
		case 'v': verbose = 1; break;
		default: helped = true; break;
		}
	}
	auto finish = [&](int st) {
		if (out_text)
			*out_text = out.release_malloc(nullptr);
		if (err_text)
			*err_text = err.release_malloc(nullptr);
		return st;
	};
	if (helped) {
		out.s += kHelp;
		return finish(0);
	}
	if (verbose)
		out.s += "verbose flag is set\n";
	if (verb == 'c') {
		pgx_tax_create(dir);
		return finish(0);
	}
	if (!verb) {
		out.s += kHelp;
		return finish(0);
	}
	pgx_taxdb *db = tax_load(dir);
	auto lookup_failed = [&](int rc) {
		err.s += rc == -1 ? "fopen: No such file or directory\n" : "fseek: Invalid argument\n";
		out.s += "Error.\n";
		status = 255;
	};
	pgx_node node;
	if (verb == 's' || verb == 'g') {
		int rc = !db->have_gi ? -1 : ((long)index - 1 < 0 ? -2 : 0);
		if (rc < 0) {
			lookup_failed(rc);
		} else {
			tax_id = (size_t)index - 1 < db->gi2tax.size() ? db->gi2tax[(size_t)index - 1] : 0;
			if (verbose)
				out.printf("%d\t%d\n", index, tax_id);
			if (tax_id == 0) {
				out.s += "0\n";
			} else if (verb == 'g') {
				rc = tax_node_record(db, tax_id, &node);
				if (rc < 0) {
					lookup_failed(rc);
				} else {
					if (verbose)
						out.printf("%d\n", node.tax_id);
					format_node_text(&node, out);
				}
			} else {
				while (tax_id != 1) {
					rc = tax_node_record(db, tax_id, &node);
					if (rc < 0) {
						lookup_failed(rc);
						break;
					}
					if (verbose)
						out.printf("%d\n", node.tax_id);
					tax_id = node.parent_tax_id;
					if (tax_id != 1)
						format_node_text(&node, out);
				}
			}
		}
	} else if (verb == 't') {
		int rc = tax_node_record(db, tax_id, &node);
		if (rc < 0) {
			lookup_failed(rc);
		} else {
			if (verbose)
				out.printf("%d\n", node.tax_id);
			format_node_text(&node, out);
		}
	} else if (verb == 'n') {
		std::vector<const uint8_t *> recs;
		int failed_seeks = 0;
		int rc = tax_names_lookup(db, tax_id, recs, &failed_seeks);
		for (int k = 0; k < failed_seeks; k++)
			err.s += k == 0 ? "fseek: Invalid argument\n" : "fseek: Bad file descriptor\n";
		if (rc < 0)
			lookup_failed(-1);
		else if (rc == 0)
			out.s += "0\n";
		else
			for (const uint8_t *r : recs)
				format_name_text(r, out);
	}
	delete db;
	return finish(status);
}

int pgx_tax_lineage_batch(pgx_taxdb *db, const int32_t *gi, int64_t n, int32_t *lineage, int32_t *count, int32_t *status)
{
	if (!db || !gi || !lineage || !count || !status || n < 0)
		return fail(PGX_E_ARG, "pgx_tax_lineage_batch: bad argument");
	PGX_TRY(require_device());
	DevBuf<int32_t> d_gi, d_lin, d_cnt, d_st;
	PGX_TRY(d_gi.alloc((size_t)n));
	PGX_TRY(d_lin.alloc((size_t)n * PGX_LINEAGE_SLOTS, 0, 0, true));
	PGX_TRY(d_cnt.alloc((size_t)n));
	PGX_TRY(d_st.alloc((size_t)n));
	PGX_TRY(d_gi.upload(gi, (size_t)n));
	PGX_TRY(tax_walk_device(db, d_gi.data(), n, d_lin.data(), d_cnt.data(), d_st.data(), nullptr));
	PGX_TRY(d_lin.download(lineage, (size_t)n * PGX_LINEAGE_SLOTS));
	PGX_TRY(d_cnt.download(count, (size_t)n));
	PGX_TRY(d_st.download(status, (size_t)n));
	return 0;
}
}
