// The RDP import on the host cores: the five-tab text of Consensus_BLAST_SOAP_RDP-1.1.pl:126-132 and the script's cursor rule
// (:141, :211, :216-220) over a whole file, all threads.  pgx_rdp_from_file (annotate.hip) takes this form for batches whose
// names are not resident in HBM, for names that repeat inside a batch and on request (PGX_RDP_HOST=1); batches made from a
// file are parsed on the device (rdp_device.hip).
//
// This header is free of HIP: it also builds with plain g++, which is how tests/host/rdp_host_test.cpp runs the threaded
// passes under AddressSanitizer + UndefinedBehaviorSanitizer and under ThreadSanitizer in the build container
// (`make -C tests/host`; tests/test_host_text.py).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace pgx {

inline uint64_t fnv64_bytes(const char *p, size_t n)
{
	uint64_t h = 1469598103934665603ull;
	for (size_t i = 0; i < n; i++)
		h = (h ^ (unsigned char)p[i]) * 1099511628211ull;
	return h;
}

inline int8_t rdp_rank_index(const std::string &t)
{
	static const char *const r[7] = { "domain", "phylum", "class", "order", "family", "genus", "species" };
	for (int i = 0; i < 7; i++)
		if (t == r[i])
			return (int8_t)i;
	return -1;
}

// Consensus:159-160: quotes and backslashes, then every [\W\d_] removed: ASCII letters remain
inline std::string clean_rdp_name(const std::string &s)
{
	std::string o;
	for (char c : s)
		if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z'))
			o += c;
	return o;
}

// Reads by name.  `Names`: size() and span(i, tmp, &p, &len) = the bytes of name i (tmp: scratch for names made on demand).
template <class Names> struct NameIndexT {
	const Names &names;
	std::vector<uint32_t> slot, next; // slot: first read of a hash (+1, 0 = empty); next: following read of the same hash (+1)
	std::vector<uint64_t> hash;
	uint64_t mask = 0;
	bool unique = true; // no two reads share a name
	explicit NameIndexT(const Names &nm) : names(nm)
	{
		const size_t n = names.size();
		size_t cap = 16;
		while (cap < 2 * n + 1)
			cap <<= 1;
		mask = cap - 1;
		slot.assign(cap, 0);
		next.assign(n, 0);
		hash.resize(n);
		// All host cores (a 2 M-read batch took 0.08 s on one): the hashes, then the table by compare-and-swap on its slots.
		// Names that do not repeat need no `next` chain, and any insertion order serves find(); the first repeated name
		// (or two names with one 64-bit hash) ends the parallel build and the table is made again in read order.
		const unsigned hw = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), n / 65536 + 1));
		auto parallel = [&](const std::function<void(size_t, size_t)> &f) {
			std::vector<std::thread> th;
			for (unsigned t = 0; t < hw; t++)
				th.emplace_back(f, n * t / hw, n * (t + 1) / hw);
			for (auto &x : th)
				x.join();
		};
		parallel([&](size_t i0, size_t i1) {
			std::string tmp;
			for (size_t i = i0; i < i1; i++) {
				const char *p;
				size_t len;
				names.span(i, tmp, &p, &len);
				hash[i] = fnv64_bytes(p, len);
			}
		});
		std::atomic<bool> repeated(false);
		uint32_t *slots = slot.data();
		parallel([&](size_t i0, size_t i1) {
			for (size_t i = i0; i < i1 && !repeated.load(std::memory_order_relaxed); i++) {
				const uint64_t h = hash[i];
				size_t k = (size_t)(h & mask);
				for (;;) {
					uint32_t cur = __atomic_load_n(&slots[k], __ATOMIC_RELAXED);
					if (cur == 0) {
						if (__atomic_compare_exchange_n(&slots[k], &cur, (uint32_t)i + 1, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED))
							break;
					}
					if (hash[cur - 1] == h) {
						repeated.store(true, std::memory_order_relaxed);
						break;
					}
					k = (k + 1) & mask;
				}
			}
		});
		if (!repeated.load())
			return;
		unique = false;
		slot.assign(cap, 0);
		std::vector<uint32_t> last(cap, 0);
		for (size_t i = 0; i < n; i++) {
			const uint64_t h = hash[i];
			size_t k = (size_t)(h & mask);
			while (slot[k] && hash[slot[k] - 1] != h)
				k = (k + 1) & mask;
			if (!slot[k])
				slot[k] = (uint32_t)i + 1;
			else
				next[last[k] - 1] = (uint32_t)i + 1;
			last[k] = (uint32_t)i + 1;
		}
	}
	size_t find(const char *text, size_t len, size_t from) const
	{
		const size_t n = names.size();
		if (n == 0)
			return n;
		const uint64_t h = fnv64_bytes(text, len);
		size_t k = (size_t)(h & mask);
		while (slot[k] && hash[slot[k] - 1] != h)
			k = (k + 1) & mask;
		std::string tmp;
		for (uint32_t i = slot[k]; i; i = next[i - 1]) {
			if ((size_t)(i - 1) < from)
				continue;
			const char *p;
			size_t l;
			names.span(i - 1, tmp, &p, &l);
			if (l == len && memcmp(p, text, len) == 0)
				return i - 1;
		}
		return n;
	}
};

struct TextRef {
	const char *p = nullptr;
	size_t n = 0;
	const char *data() const { return p; }
	size_t size() const { return n; }
	bool empty() const { return n == 0; }
	char back() const { return p[n - 1]; }
};

// what the host import makes, and the work arrays it leaves for its caller to free (a few hundred megabytes for a 2 M-line file)
struct RdpHostTable {
	std::vector<uint32_t> trips;  // per read: triplets of its line
	std::vector<uint8_t> present; // per read: it has a line
	size_t n_trip = 0;
	std::unique_ptr<uint32_t[]> name_a, code_a; // per triplet, in read order: token id of the cleaned name; name << 3 | rank + 1
	std::unique_ptr<int8_t[]> rank_a;           // index in (domain .. species) or -1
	// work arrays
	std::vector<size_t> ls;
	std::vector<uint32_t> line_read, id_len;
	std::vector<uint8_t> has_five;
	std::vector<std::vector<uint32_t>> t_name, tok_of;
	std::vector<std::vector<int8_t>> t_rank;
	std::vector<std::vector<std::string>> t_local;
};

// `index`: a NameIndexT over the batch's names; `hw_env`: threads (0 = up to 16 by the file's size); `intern`: cleaned name ->
// token id (called from ONE thread, after the parallel passes); `lap`: called with a stage's name after each stage
template <class Index, class Intern, class Lap>
void rdp_parse_host(const TextRef text, size_t n, const Index &index, unsigned hw_env, Intern &&intern, Lap &&lap, RdpHostTable &out)
{
	std::vector<uint32_t> &trips = out.trips;
	std::vector<uint8_t> &present = out.present;
	trips.assign(n ? n : 1, 0);
	present.assign(n ? n : 1, 0);
	std::vector<size_t> &ls = out.ls;
	std::vector<uint32_t> &line_read = out.line_read, &id_len = out.id_len;
	std::vector<uint8_t> &has_five = out.has_five;
	std::vector<std::vector<uint32_t>> &t_name = out.t_name, &tok_of = out.tok_of;
	std::vector<std::vector<int8_t>> &t_rank = out.t_rank;
	std::vector<std::vector<std::string>> &t_local = out.t_local;
	std::unique_ptr<uint32_t[]> &name_a = out.name_a, &code_a = out.code_a;
	std::unique_ptr<int8_t[]> &rank_a = out.rank_a;
	size_t &n_trip = out.n_trip;
	lap("name index");
	static const char kFive[] = "\t\t\t\t\t";
	const char *base = text.data();
	// ---- lines (all host cores: each takes a stretch of the text and notes the byte after every newline in it)
	{
		const unsigned tw = (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), text.size() / (1u << 20) + 1));
		std::vector<std::vector<size_t>> part(tw);
		std::vector<std::thread> th;
		for (unsigned t = 0; t < tw; t++)
			th.emplace_back([&, t]() {
				const size_t c0 = text.size() * t / tw, c1 = text.size() * (t + 1) / tw;
				std::vector<size_t> mine; // (the thread's own vector: see pass 2)
				mine.reserve((c1 - c0) / 64 + 16);
				for (size_t s0 = c0; s0 < c1;) {
					const char *nl = (const char *)memchr(base + s0, '\n', c1 - s0);
					if (!nl)
						break;
					s0 = (size_t)(nl - base) + 1;
					if (s0 < text.size())
						mine.push_back(s0);
				}
				part[t] = std::move(mine);
			});
		for (auto &x : th)
			x.join();
		size_t total = text.empty() ? 0 : 1;
		for (auto &v : part)
			total += v.size();
		ls.reserve(total + 1);
		if (!text.empty())
			ls.push_back(0);
		for (auto &v : part)
			ls.insert(ls.end(), v.begin(), v.end());
	}
	const size_t n_lines = ls.size();
	lap("line starts");
	ls.push_back(text.size() + (text.empty() || text.back() != '\n' ? 1 : 0)); // (line i ends one byte before the next start)
	auto line_of = [&](size_t i, const char **line, size_t *len) {
		*line = base + ls[i];
		const size_t e = i + 1 < n_lines ? ls[i + 1] - 1 : (text.size() && text.back() == '\n' ? text.size() - 1 : text.size());
		*len = e - ls[i];
	};
	const unsigned hw = hw_env ? hw_env : (unsigned)std::max<size_t>(1, std::min<size_t>(std::min(16u, std::max(1u, std::thread::hardware_concurrency())), n_lines / 4096 + 1));
	auto parallel = [&](const std::function<void(unsigned, size_t, size_t)> &f) {
		std::vector<std::thread> th;
		for (unsigned t = 0; t < hw; t++)
			th.emplace_back(f, t, n_lines * t / hw, n_lines * (t + 1) / hw);
		for (auto &x : th)
			x.join();
	};
	// ---- pass 1 (all host cores): the read each line names.  With names that do not repeat inside the batch the cursor
	// rule "first read at or after the cursor" is "the one read of that name, if it is not behind the cursor": a probe per
	// line, independent of the others; batches with repeated names keep the sequential walk.
	line_read.assign(n_lines, (uint32_t)n);
	id_len.assign(n_lines, 0);
	has_five.assign(n_lines, 0);
	parallel([&](unsigned, size_t i0, size_t i1) {
		for (size_t i = i0; i < i1; i++) {
			const char *line;
			size_t len;
			line_of(i, &line, &len);
			// (the first run of five tabs: from tab to tab -- the C library's memmem spends most of a 130-byte line setting up)
			const char *five = nullptr;
			for (const char *tb = (const char *)memchr(line, '\t', len); tb && (size_t)(tb - line) + 5 <= len;) {
				size_t run = 1;
				while (run < 5 && tb[run] == '\t')
					run++;
				if (run == 5) {
					five = tb;
					break;
				}
				tb = (const char *)memchr(tb + run, '\t', len - (size_t)(tb + run - line));
			}
			id_len[i] = (uint32_t)(five ? (size_t)(five - line) : len);
			has_five[i] = five != nullptr;
			if (index.unique)
				line_read[i] = (uint32_t)index.find(line, id_len[i], 0);
		}
	});
	lap("pass 1 (reads of the lines)");
	{
		size_t cursor = 0;
		for (size_t i = 0; i < n_lines; i++) {
			size_t r = line_read[i];
			if (!index.unique)
				r = index.find(base + ls[i], id_len[i], cursor);
			else if (r < cursor)
				r = n; // its only read lies behind the cursor
			line_read[i] = (uint32_t)r;
			if (r < n) {
				cursor = r + 1;
				present[r] = 1;
			}
		}
	}
	// ---- pass 2 (all host cores): the (name, rank, confidence) fields of the lines that belong to a read.  The few
	// distinct name / rank texts of an RDP file are cleaned and interned once each (per thread; the database's token table
	// behind a lock)
	lap("cursor rule");
	// per-thread memo of the distinct field texts (a few thousand per file): open addressing on the text's hash, the text
	// itself compared on a hit; a full table (never seen) just stops remembering
	struct Memo {
		struct Slot {
			uint64_t h = 0;
			const char *p = nullptr;
			uint32_t len = 0, value = 0;
		};
		std::vector<Slot> slot;
		size_t used = 0;
		explicit Memo(size_t slots) : slot(slots) {}
		// the slot of the text: *hit says whether it already holds a value.  The table starts small (an RDP file names a few
		// thousand distinct taxa: 96 KB stay in the core's cache -- the fixed 6 MB table of round 2 missed on every field, 0.28 s
		// of the 0.56 s a 2 M-line file took) and grows fourfold when half full, up to 2^20 slots
		Slot *find(const char *p, size_t len, bool *hit) { return find_h(p, len, fnv64_bytes(p, len), hit); }
		Slot *find_h(const char *p, size_t len, uint64_t hash, bool *hit) // hash = fnv64_bytes(p, len)
		{
			const uint64_t h = hash | 1ull; // 0 marks an empty slot
			for (;;) {
				size_t k = (size_t)(h >> 8) & (slot.size() - 1);
				for (;;) {
					Slot &e = slot[k];
					if (e.h == h && e.len == len && memcmp(e.p, p, len) == 0) {
						*hit = true;
						return &e;
					}
					if (e.h == 0)
						break;
					k = (k + 1) & (slot.size() - 1);
				}
				*hit = false;
				if (2 * (used + 1) > slot.size()) {
					if (slot.size() >= (1u << 20))
						return nullptr;
					std::vector<Slot> old(slot.size() * 4);
					old.swap(slot);
					for (const Slot &o : old) {
						if (!o.h)
							continue;
						size_t j = (size_t)(o.h >> 8) & (slot.size() - 1);
						while (slot[j].h)
							j = (j + 1) & (slot.size() - 1);
						slot[j] = o;
					}
					continue; // (probe the grown table)
				}
				Slot &e = slot[k];
				e.h = h;
				e.p = p; // (the file's text outlives the memo)
				e.len = (uint32_t)len;
				used++;
				return &e;
			}
		}
	};
	// (the database's token table is touched after the parallel part, once per distinct cleaned name of a thread: with the
	// table behind a lock, the first sight of 40 000 names in each of 16 threads cost 0.3 s of a 0.6 s pass)
	t_name.assign(hw, {});
	t_rank.assign(hw, {});
	t_local.assign(hw, {}); // per thread: its distinct cleaned names, by local number
	parallel([&](unsigned t, size_t i0, size_t i1) {
		std::unordered_map<std::string, uint32_t> lmap;
		std::vector<std::string> loc;
		// (names: a database with 33 000 genera filled a 65 536-slot table half way, after which every field took the
		// interning lock: 0.9 s instead of 0.5 s for 2 M lines)
		Memo names(1 << 12), ranks(1 << 8);
		// (the thread's OWN vectors, moved into t_name / t_rank / t_local at the end: the headers of those sixteen vectors lie
		// side by side in memory, and every push_back through a reference to one of them wrote its end pointer into a cache
		// line that two other threads were writing theirs to -- this pass took 0.27 s on one thread, 0.60 s on four and
		// 0.28 s on sixteen)
		std::vector<uint32_t> nm;
		std::vector<int8_t> rk;
		nm.reserve((i1 - i0) * 8 + 64);
		rk.reserve((i1 - i0) * 8 + 64);
		for (size_t i = i0; i < i1; i++) {
			const size_t r = line_read[i];
			if (r >= n || !has_five[i])
				continue;
			const char *line;
			size_t len;
			line_of(i, &line, &len);
			const char *rest = line + id_len[i] + 5;
			size_t rest_len = len - id_len[i] - 5;
			// (a second five-tab group ends the fields, as before; then ONE walk over the bytes: a field ends at a tab and its
			// FNV hash is made on the way -- a memchr and a separate hash pass per field were most of this pass)
			if (const char *again = (const char *)memmem(rest, rest_len, kFive, 5))
				rest_len = (size_t)(again - rest);
			while (rest_len && rest[rest_len - 1] == '\t') // trailing empty fields are dropped
				rest_len--;
			if (rest_len == 0)
				continue;
			size_t a = 0; // start of the current field
			uint64_t fh = 1469598103934665603ull;
			int k = 0;
			for (size_t x = 0; x <= rest_len; x++) {
				if (x < rest_len && rest[x] != '\t') {
					fh = (fh ^ (unsigned char)rest[x]) * 1099511628211ull;
					continue;
				}
				// field k = rest[a, x)
				if (k % 3 == 0) {
					bool hit;
					auto *e = names.find_h(rest + a, x - a, fh, &hit);
					uint32_t tok = hit ? e->value : 0u;
					if (!hit) {
						const std::string clean = clean_rdp_name(std::string(rest + a, x - a));
						auto it = lmap.find(clean);
						if (it == lmap.end()) {
							it = lmap.emplace(clean, (uint32_t)loc.size()).first;
							loc.push_back(clean);
						}
						tok = it->second; // the thread's own number of the name; token ids follow below
						if (e)
							e->value = tok;
					}
					nm.push_back(tok);
					rk.push_back((int8_t)-1);
					trips[r]++; // (one line per read: no two threads touch one counter)
				} else if (k % 3 == 1) {
					bool hit;
					auto *e = ranks.find_h(rest + a, x - a, fh, &hit);
					uint32_t rv = hit ? e->value : (uint32_t)(uint8_t)rdp_rank_index(std::string(rest + a, x - a));
					if (!hit && e)
						e->value = rv;
					rk.back() = (int8_t)rv;
				}
				k++;
				a = x + 1;
				fh = 1469598103934665603ull;
			}
		}
		t_name[t] = std::move(nm);
		t_rank[t] = std::move(rk);
		t_local[t] = std::move(loc);
	});
	lap("pass 2 (fields)");
	// the threads' pieces behind one another (matched reads come in increasing order, so the triplets already lie in read
	// order): token ids for each thread's distinct names (the database's table, one thread), then every thread maps and
	// copies its own piece into arrays that nobody zero-filled first
	tok_of.assign(hw, {});
	std::vector<size_t> piece_at(hw + 1, 0);
	for (unsigned t = 0; t < hw; t++) {
		tok_of[t].resize(t_local[t].size());
		for (size_t k = 0; k < tok_of[t].size(); k++)
			tok_of[t][k] = intern(t_local[t][k]);
		piece_at[t + 1] = piece_at[t] + t_name[t].size();
	}
	n_trip = piece_at[hw];
	name_a.reset(new uint32_t[n_trip + 1]);
	code_a.reset(new uint32_t[n_trip + 1]);
	rank_a.reset(new int8_t[n_trip + 1]);
	parallel([&](unsigned t, size_t, size_t) {
		uint32_t *nm = name_a.get() + piece_at[t], *cd = code_a.get() + piece_at[t];
		int8_t *rk = rank_a.get() + piece_at[t];
		const std::vector<uint32_t> &src = t_name[t], &map = tok_of[t];
		const std::vector<int8_t> &srk = t_rank[t];
		for (size_t k = 0; k < src.size(); k++) {
			const uint32_t tk = map[src[k]];
			nm[k] = tk;
			rk[k] = srk[k];
			cd[k] = (tk << 3) | (uint32_t)(srk[k] + 1);
		}
	});
	lap("concatenate");
}

} // namespace pgx
