// The host form of the RDP import (pangea-plus_amd/csrc/rdp_host.hpp: the threaded passes behind pgx_rdp_from_file when a
// batch's names are not resident in HBM) built with plain g++ and run under the sanitizers in the build container -- no
// GPU, no HIP: `make -C tests/host asan tsan` (tests/test_host_text.py runs both).  The parse is compared with a sequential
// reading of the same text written independently below (the script's rules, Consensus_BLAST_SOAP_RDP-1.1.pl:126-132, :141,
// :211-220), for 1, 3, 8 and 16 threads, on files in odd shapes.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

#include "../../pangea-plus_amd/csrc/rdp_host.hpp"

using namespace pgx;

struct VecNames {
	std::vector<std::string> v;
	size_t size() const { return v.size(); }
	void span(size_t i, std::string &, const char **p, size_t *len) const
	{
		*p = v[i].data();
		*len = v[i].size();
	}
};

struct Plain { // what the sequential reading makes: per read its (cleaned name, rank index) pairs, or "no line"
	std::vector<int> present;
	std::vector<std::vector<std::pair<std::string, int>>> trip;
};

static Plain read_plain(const std::string &text, const VecNames &names)
{
	Plain out;
	out.present.assign(names.size(), 0);
	out.trip.resize(names.size());
	std::vector<std::string> lines;
	size_t s = 0;
	while (s < text.size()) {
		size_t e = text.find('\n', s);
		if (e == std::string::npos)
			e = text.size();
		lines.push_back(text.substr(s, e - s));
		s = e + 1;
	}
	size_t cursor = 0;
	for (const std::string &l : lines) {
		const size_t five = l.find("\t\t\t\t\t");
		const std::string id = five == std::string::npos ? l : l.substr(0, five);
		size_t r = names.size();
		for (size_t k = cursor; k < names.size(); k++)
			if (names.v[k] == id) {
				r = k;
				break;
			}
		if (r == names.size())
			continue;
		cursor = r + 1;
		out.present[r] = 1;
		if (five == std::string::npos)
			continue;
		std::string rest = l.substr(five + 5);
		const size_t again = rest.find("\t\t\t\t\t");
		if (again != std::string::npos)
			rest.resize(again);
		while (!rest.empty() && rest.back() == '\t')
			rest.pop_back();
		if (rest.empty())
			continue;
		std::vector<std::string> f;
		size_t a = 0;
		for (;;) {
			const size_t t = rest.find('\t', a);
			f.push_back(rest.substr(a, t == std::string::npos ? std::string::npos : t - a));
			if (t == std::string::npos)
				break;
			a = t + 1;
		}
		for (size_t k = 0; k < f.size(); k += 3)
			out.trip[r].push_back({ clean_rdp_name(f[k]), k + 1 < f.size() ? (int)rdp_rank_index(f[k + 1]) : -1 });
	}
	return out;
}

static int check(const std::string &text, const VecNames &names, const char *what)
{
	const Plain want = read_plain(text, names);
	int bad = 0;
	for (unsigned threads : { 1u, 3u, 8u, 16u }) {
		NameIndexT<VecNames> index(names);
		std::map<std::string, uint32_t> tok;
		std::vector<std::string> tok_text;
		RdpHostTable ht;
		rdp_parse_host(
			TextRef{ text.data(), text.size() }, names.size(), index, threads,
			[&](const std::string &nm) {
				auto it = tok.find(nm);
				if (it == tok.end()) {
					it = tok.emplace(nm, (uint32_t)tok_text.size()).first;
					tok_text.push_back(nm);
				}
				return it->second;
			},
			[](const char *) {}, ht);
		size_t at = 0;
		for (size_t r = 0; r < names.size(); r++) {
			if ((int)ht.present[r] != want.present[r] || ht.trips[r] != want.trip[r].size()) {
				if (bad++ < 5)
					fprintf(stderr, "%s, %u threads: read %zu present %d/%d triplets %u/%zu\n", what, threads, r, (int)ht.present[r], want.present[r],
						ht.trips[r], want.trip[r].size());
				at += ht.trips[r];
				continue;
			}
			for (size_t k = 0; k < want.trip[r].size(); k++, at++) {
				const std::string &nm = tok_text[ht.name_a[at]];
				if (nm != want.trip[r][k].first || (int)ht.rank_a[at] != want.trip[r][k].second ||
				    ht.code_a[at] != ((ht.name_a[at] << 3) | (uint32_t)(ht.rank_a[at] + 1))) {
					if (bad++ < 5)
						fprintf(stderr, "%s, %u threads: read %zu triplet %zu: %s/%d, want %s/%d\n", what, threads, r, k, nm.c_str(), (int)ht.rank_a[at],
							want.trip[r][k].first.c_str(), want.trip[r][k].second);
				}
			}
		}
		if (at != ht.n_trip)
			bad++;
	}
	printf("%-28s %s\n", what, bad ? "DIFFERENT" : "ok");
	return bad;
}

int main()
{
	std::mt19937_64 rng(20261005);
	auto pick = [&](size_t n) { return (size_t)(rng() % n); };
	const char *taxa[] = { "Bacteria", "\"Proteobacteria\"", "Gamma proteobacteria 7", "Escherichia/Shigella", "unclassified_\"x\"", "", "Archaea", "Incertae Sedis XI" };
	const char *ranks[] = { "domain", "phylum", "class", "order", "family", "genus", "species", "norank", "" };
	int bad = 0;
	for (int round = 0; round < 6; round++) {
		VecNames names;
		const size_t n = round == 5 ? 70000 : 300 + 517 * (size_t)round;
		for (size_t i = 0; i < n; i++)
			names.v.push_back("r" + std::to_string(i * 7 + (size_t)round));
		if (round == 3) { // names that repeat inside the batch: the chains of the index, the sequential cursor walk
			for (size_t i = 5; i < n; i += 11)
				names.v[i] = names.v[i - 3];
		}
		std::string text;
		for (size_t i = 0; i < n; i++) {
			const size_t shape = pick(14);
			if (shape == 0)
				continue; // a read without a line
			std::string l = names.v[i];
			if (shape == 1) {
				text += l + "\n"; // the id alone
				continue;
			}
			l += shape == 2 ? "\t\t\t\t\t\t\t" : "\t\t\t\t\t";
			const size_t nf = 1 + pick(9);
			for (size_t k = 0; k < nf; k++) {
				if (k)
					l += '\t';
				l += k % 3 == 0 ? taxa[pick(8)] : (k % 3 == 1 ? ranks[pick(9)] : "0.93");
			}
			if (shape == 3)
				l += "\t\t\t\t\tBacteria\tdomain\t1";
			if (shape == 4)
				l += "\t\t\t";
			if (shape == 5)
				text += "nobody" + std::to_string(i) + "\t\t\t\t\tBacteria\tdomain\t1\n";
			if (shape == 6 && i > 20)
				text += names.v[i - 17] + "\t\t\t\t\tArchaea\tdomain\t1\n"; // a read behind the cursor
			if (shape == 7)
				text += "\n";
			if (shape == 8)
				l += "\r";
			text += l + "\n";
		}
		if (round % 2)
			text.pop_back(); // no newline at the very end
		char what[64];
		snprintf(what, sizeof what, "round %d (%zu reads)", round, n);
		bad += check(text, names, what);
	}
	{
		VecNames names;
		names.v = { "a", "b" };
		bad += check("", names, "empty file");
		bad += check("\n\n\n", names, "newlines only");
		bad += check("b\t\t\t\t\tX\tgenus\t1\na\t\t\t\t\tY\tgenus\t1", names, "second read first");
	}
	return bad ? 1 : 0;
}
