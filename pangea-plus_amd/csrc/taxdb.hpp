// Taxonomy database object (Tax_class/ncbitc.c as a library) and helpers shared with the
// tax-annotate and consensus stages.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "common.hpp"

struct pgx_taxdb {
	bool have_gi = false, have_nodes = false, have_names = false;
	std::vector<int32_t> gi2tax;   // gi_taxid_nucl.dmp.bin
	std::vector<uint8_t> nodes;    // nodes.dmp.bin, 28-byte records
	size_t n_nodes = 0;
	std::vector<uint8_t> names;    // names.dmp.bin: int32 count + 196-byte records
	int32_t n_names = 0;
	size_t names_records = 0;
	// HBM: what the lineage walk needs
	pgx::DevBuf<int32_t> d_gi2tax, d_node_taxid, d_node_parent;
	pgx::DevBuf<int8_t> d_node_code;
};

namespace pgx {
const char *tax_rank_text(int id);
int8_t driver_rank_code(int rank_enum);
int tax_node_record(const pgx_taxdb *db, int taxid, pgx_node *out);
int tax_names_lookup(const pgx_taxdb *db, int taxid, std::vector<const uint8_t *> &out, int *failed_seeks = nullptr);
bool tax_scientific_name(const pgx_taxdb *db, int taxid, std::string &name);
int tax_walk_device(pgx_taxdb *db, const int32_t *d_gi, int64_t n, int32_t *d_lineage, int32_t *d_count,
		    int32_t *d_status, int32_t *d_leaf);

// Lineage text exactly as NCBI-taxcollector-0.01.pl:96-144 prints it, from the elements the walk
// kept (leaf first). `gi_text` is used for the "Unidentified(GI:n);" element (status 1).
struct LineageRenderer {
	const pgx_taxdb *db;
	explicit LineageRenderer(const pgx_taxdb *d) : db(d) {}
	std::string render(const int32_t *lineage, int count, int status, const std::string &gi_text);
	// per-taxid cache of "[idx]" + "name;|" pieces
	std::vector<std::pair<int32_t, std::string>> cache_keys;
	std::string piece(int32_t taxid);
	std::unordered_map<int32_t, std::string> *cache = nullptr;
	~LineageRenderer();
};
} // namespace pgx
