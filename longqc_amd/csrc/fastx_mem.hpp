// longqc_amd/csrc/fastx_mem.hpp -- FASTA/FASTQ records of a plain (not gzip) file, parsed from memory by many threads.
//
// Same record grammar as fastx.hpp's streaming reader, i.e. what the reference accepts (kseq.h:179-224 via bseq.c:56-102):
//   * a record starts at the next '>' or '@' (anywhere, when no header character is pending); the name is the header up to
//     the first whitespace, the rest of the line is a comment;
//   * sequence lines are concatenated until a line that starts with '>', '@' or '+'; a '\r' that ends the sequence so far is
//     dropped after every line; after '+' the rest of that line is skipped and quality lines are read until they cover the
//     sequence; a quality string of another length, or none, ends the stream.
// The reference reads its input with one thread (kseq over gzread, bseq.c:68-102) and the engine's own streaming reader
// did the same: 614 Mbases/s end to end in round 1, six times below what the GPU side takes.  Here the file is mapped
// and cut into pieces; every piece is parsed speculatively from a guessed record start (FASTQ: a line that starts with
// '@' whose next-but-one line starts with '+' and whose fourth line is as long as its second; FASTA: a line that starts
// with '>'), pieces are then stitched in file order: piece k + 1 is accepted only if piece k, parsed to its end, stops
// exactly at that guess with no header character pending -- otherwise it is parsed again from where piece k really ended
// (wrapped FASTQ, '@' lines that fool the guess: correct, just not parallel).  The records are descriptors into the
// mapping (sequences of one line are not copied); 2-bit packing into the engine's layout (lq_pack_host's) happens in
// parallel, read by read, straight into the buffers that are uploaded.
#pragma once
#include "lq_common.hpp"
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <cstring>
#include <stdexcept>
#include <algorithm>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

struct MemRec {
	u64 name_off; u32 name_len;      // into the mapping
	u64 seq_off; u32 seq_len;        // into the mapping (own == 0) or into the piece's side buffer (own == 1: several lines / a dropped '\r')
	u32 own;
};

struct MemPiece {
	u64 begin = 0, end = 0;          // parsed [begin, ...): records whose header character lies before `end`
	u64 stop = 0;                    // where the parser stood after its last record (the next header character, if pending, is at stop - 1)
	int last_char = 0;               // header character pending at `stop` (kseq's last_char)
	bool stream_over = false;        // a truncated quality string: the stream ends here (kseq returns -2)
	std::vector<MemRec> recs;
	std::vector<u8> side;
};

class MemFastx {
	const u8 *p_ = nullptr; u64 n_ = 0; int fd_ = -1; bool mapped_ = false;
	std::vector<u8> owned_;
public:
	MemFastx() {}
	~MemFastx() { close(); }
	MemFastx(const MemFastx&) = delete;
	const u8 *data() const { return p_; }
	u64 size() const { return n_; }
	void close()
	{
		if (mapped_ && p_) munmap((void*)p_, n_);
		if (fd_ >= 0) ::close(fd_);
		p_ = nullptr; n_ = 0; fd_ = -1; mapped_ = false; owned_.clear();
	}
	// true: the file is mapped and is not gzip; false: not a regular plain file (the caller uses the streaming reader)
	bool open(const std::string &path)
	{
		close();
		fd_ = ::open(path.c_str(), O_RDONLY);
		if (fd_ < 0) throw std::runtime_error("failed to open file '" + path + "'");
		struct stat st;
		if (fstat(fd_, &st) != 0 || !S_ISREG(st.st_mode)) { ::close(fd_); fd_ = -1; return false; }
		n_ = (u64)st.st_size;
		if (n_ == 0) { mapped_ = false; p_ = (const u8*)""; return true; }
		void *m = mmap(nullptr, n_, PROT_READ, MAP_PRIVATE, fd_, 0);
		if (m == MAP_FAILED) { ::close(fd_); fd_ = -1; n_ = 0; return false; }
		p_ = (const u8*)m; mapped_ = true;
		madvise(m, n_, MADV_SEQUENTIAL);
		if (n_ >= 2 && p_[0] == 0x1f && p_[1] == 0x8b) { close(); return false; }      // gzip: one stream, one inflater
		return true;
	}

	static inline bool is_space(u8 c) { return c == ' ' || (c >= '\t' && c <= '\r'); }   // isspace in the C locale

	// the first '>' or '@' in [pos, n), or null.  After a FASTQ record the very next byte is the header; elsewhere the search
	// runs over growing windows, both characters per window: a file without any '>' (or '@') must not be scanned to its end
	// once per record (round 4's two whole-file memchr calls made an 80-MB FASTQ take 40 s).
	static inline const u8 *next_header(const u8 *p, u64 pos, u64 n)
	{
		if (pos < n && (p[pos] == '>' || p[pos] == '@')) return p + pos;
		for (u64 win = 256; pos < n; win = win < (1u << 20) ? win * 4 : win) {
			const u64 len = std::min<u64>(win, n - pos);
			const u8 *a = (const u8*)memchr(p + pos, '>', len), *b = (const u8*)memchr(p + pos, '@', a ? (size_t)(a - (p + pos)) : len);
			if (b) return b;
			if (a) return a;
			pos += len;
		}
		return nullptr;
	}

	// kseq's record loop from position `pos` with `last_char` pending, for records whose header character lies before `end`
	// (end == size(): to the end of the file).  keep_qual is not needed here (targets only).
	void parse(u64 pos, int last_char, u64 end, MemPiece &out) const
	{
		const u8 *p = p_; const u64 n = n_;
		out.recs.clear(); out.side.clear(); out.stream_over = false;
		for (;;) {
			if (last_char == 0) {                                   // to the next header character
				const u8 *h = next_header(p, pos, n);
				if (!h) { pos = n; break; }
				if ((u64)(h - p) >= end) { pos = (u64)(h - p); break; }      // belongs to the next piece: stop in front of it
				pos = (u64)(h - p) + 1; last_char = *h;
			} else if (pos - 1 >= end) { break; }                      // the pending header character belongs to the next piece
			// name: up to the first whitespace; the rest of the line is a comment
			if (pos >= n) { last_char = 0; pos = n; break; }           // the stream ended right after a header character: no record (kseq: -1)
			MemRec r; r.name_off = pos; r.own = 0;
			u64 q = pos;
			while (q < n && !is_space(p[q])) ++q;
			r.name_len = (u32)(q - pos);
			if (q < n && p[q] != '\n') { const u8 *e = (const u8*)memchr(p + q, '\n', n - q); q = e ? (u64)(e - p) : n; }
			pos = q < n ? q + 1 : n;
			// sequence lines until a line that starts with '>', '@' or '+'
			u64 s_off = 0, s_len = 0; bool first = true, own = false; size_t side0 = out.side.size();
			int c = -1;
			for (;;) {
				if (pos >= n) { c = -1; break; }
				c = p[pos++];
				if (c == '>' || c == '+' || c == '@') break;
				if (c == '\n') continue;
				const u64 l0 = pos - 1;
				const u8 *e = (const u8*)memchr(p + pos, '\n', n - pos);
				u64 l1 = e ? (u64)(e - p) : n;                          // line = [l0, l1)
				pos = e ? l1 + 1 : n;
				// until_: append the line, then drop one trailing '\r' of the sequence so far (if it is longer than one character)
				if (first) { s_off = l0; s_len = l1 - l0; first = false; if (s_len > 1 && p[l1 - 1] == '\r') --s_len; }
				else {
					if (!own) { out.side.insert(out.side.end(), p + s_off, p + s_off + s_len); own = true; }
					out.side.insert(out.side.end(), p + l0, p + l1);
					// (a line that is the file's last byte: kseq's ks_getuntil2 returns at end of file before it looks for the '\r', kseq.h:98.
					// One case stays outside the parity domain, INTEGRATION.md: a file of a whole number of kseq's 16384-byte buffers -- its stream
					// learns of the end one call later and drops that '\r'; neither reader here counts buffers)
					if (out.side.size() - side0 > 1 && out.side.back() == '\r' && !(l0 + 1 == n)) out.side.pop_back();
				}
			}
			if (own) { r.seq_off = side0; r.seq_len = (u32)(out.side.size() - side0); r.own = 1; if (out.side.size() - side0 > 0x7fffffffULL) throw std::domain_error("read longer than 2^31-1 bases (bseq.c:80)"); }
			else { r.seq_off = s_off; r.seq_len = (u32)s_len; if (s_len > 0x7fffffffULL) throw std::domain_error("read longer than 2^31-1 bases (bseq.c:80)"); }
			if (c == '>' || c == '@') { last_char = c; out.recs.push_back(r); continue; }
			if (c != '+') { last_char = 0; out.recs.push_back(r); pos = n; break; }     // end of file: the last record (FASTA)
			// '+': skip the rest of the line, then quality lines until they cover the sequence
			{
				const u8 *e = pos < n ? (const u8*)memchr(p + pos, '\n', n - pos) : nullptr;
				if (!e) { out.stream_over = true; last_char = 0; pos = n; break; }        // no quality string: kseq returns -2
				pos = (u64)(e - p) + 1;
			}
			// kseq appends every quality line to the string so far and then drops ONE trailing '\r' of that whole string if it is
			// longer than one character (kseq.h:98-99, append mode) -- so an empty line after a line that ended in "\r\r" drops
			// the second one.  Only the length matters here: qlen and the number of '\r' the string so far ends with.
			u64 qlen = 0, trail = 0;
			const u64 want = own ? r.seq_len : s_len;
			for (;;) {                                                  // while (until_(qual) && qual.size() < seq.size())
				if (pos >= n) break;
				const u8 *e = (const u8*)memchr(p + pos, '\n', n - pos);
				const u64 l1 = e ? (u64)(e - p) : n;
				u64 t = 0;
				while (t < l1 - pos && p[l1 - 1 - t] == '\r') ++t;
				trail = (t == l1 - pos) ? trail + t : t;
				qlen += l1 - pos;
				if (qlen > 1 && trail > 0) { --qlen; --trail; }
				pos = e ? l1 + 1 : n;
				if (qlen >= want) break;
			}
			last_char = 0;
			if (qlen != want) { out.stream_over = true; break; }         // truncated quality: kseq returns -2, the stream ends
			out.recs.push_back(r);
		}
		out.stop = pos; out.last_char = last_char;
	}

	// a guessed record start at or after `from` (a position where a parser with no header character pending would find its next
	// header), or size() if none: see the file comment
	u64 guess_start(u64 from, bool fastq) const
	{
		const u8 *p = p_; const u64 n = n_;
		u64 q = from;
		if (q > 0) { const u8 *e = (const u8*)memchr(p + q - 1, '\n', n - (q - 1)); if (!e) return n; q = (u64)(e - p) + 1; }   // start of a line
		for (int tries = 0; q < n && tries < 64; ++tries) {
			const u8 *e1 = (const u8*)memchr(p + q, '\n', n - q);
			if (!fastq) { if (p[q] == '>') return q; }
			else if (p[q] == '@' && e1) {
				const u64 l2 = (u64)(e1 - p) + 1;                      // sequence line
				const u8 *e2 = l2 < n ? (const u8*)memchr(p + l2, '\n', n - l2) : nullptr;
				if (e2) {
					const u64 l3 = (u64)(e2 - p) + 1;                  // '+' line
					const u8 *e3 = l3 < n ? (const u8*)memchr(p + l3, '\n', n - l3) : nullptr;
					if (e3 && p[l3] == '+' && p[l2] != '@' && p[l2] != '+' && p[l2] != '>') {
						const u64 l4 = (u64)(e3 - p) + 1;              // quality line
						const u8 *e4 = l4 < n ? (const u8*)memchr(p + l4, '\n', n - l4) : nullptr;
						const u64 ql = (e4 ? (u64)(e4 - p) : n) - l4, sl = (u64)(e2 - p) - l2;
						if (ql == sl) return q;
					}
				}
			}
			if (!e1) return n;
			q = (u64)(e1 - p) + 1;
		}
		return q < n ? q : n;                                           // (no convincing start nearby: any line start; the stitcher decides)
	}
};

// All records of the file, in order, parsed by `n_threads` threads over pieces of about `piece_bytes`.
struct MemRecords {
	std::vector<MemPiece> pieces;          // in file order; a record's bytes: own ? pieces[k].side : the mapping
	u64 n_recs = 0;
	u32 reparsed = 0;                      // pieces whose guessed start did not hold (parsed again in order)
	bool too_wrapped = false;              // records of several lines are copied (MemPiece::side): past side_limit bytes of such copies the parse stops and the caller streams the file instead
};

inline void lq_parse_all(const MemFastx &f, int n_threads, u64 piece_bytes, MemRecords &out, u64 side_limit = ~0ULL)
{
	const u64 n = f.size();
	out.pieces.clear(); out.n_recs = 0; out.reparsed = 0; out.too_wrapped = false;
	if (n == 0) return;
	if (piece_bytes < 64) piece_bytes = 64;
	if (n_threads <= 0) n_threads = (int)std::min<unsigned>(64, std::max(1u, std::thread::hardware_concurrency()));
	// the first header character decides which guess is used
	bool fastq = true;
	{ const u8 *h = MemFastx::next_header(f.data(), 0, n); fastq = h && *h == '@'; }
	std::vector<u64> starts{0};
	for (u64 at = piece_bytes; at < n; at += piece_bytes) {
		const u64 g = f.guess_start(at, fastq);
		if (g < n && g > starts.back()) starts.push_back(g);
	}
	const size_t np = starts.size();
	out.pieces.resize(np);
	std::atomic<size_t> next(0);
	std::atomic<u64> side_bytes(0);
	std::atomic<bool> stop(false);
	std::vector<std::exception_ptr> errs((size_t)n_threads);
	auto work = [&](int ti) {
		try {
			for (;;) {
				const size_t k = next.fetch_add(1);
				if (k >= np || stop.load()) break;
				MemPiece &pc = out.pieces[k];
				pc.begin = starts[k]; pc.end = k + 1 < np ? starts[k + 1] : n;
				f.parse(pc.begin, 0, pc.end, pc);
				if (side_bytes.fetch_add(pc.side.size()) + pc.side.size() > side_limit) stop.store(true);
			}
		} catch (...) { errs[(size_t)ti] = std::current_exception(); }
	};
	if (n_threads == 1 || np == 1) work(0);
	else { std::vector<std::thread> th; for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t); for (auto &t : th) t.join(); }
	for (auto &e : errs) if (e) std::rethrow_exception(e);
	if (stop.load()) { out.pieces.clear(); out.too_wrapped = true; return; }
	// stitch: piece k + 1 holds iff piece k stopped exactly in front of its guess with nothing pending
	for (size_t k = 0; k < np; ++k) {
		MemPiece &pc = out.pieces[k];
		if (pc.stream_over) { out.pieces.resize(k + 1); break; }
		if (k + 1 < np) {
			MemPiece &nx = out.pieces[k + 1];
			const bool holds = (pc.last_char == 0 && pc.stop == nx.begin) ||
			                   (pc.last_char != 0 && pc.stop == nx.begin + 1);      // (a FASTA record ends by reading the next header character: pending, and the guess is that character)
			if (!holds) {                                               // re-parse the next piece from where this one really ended
				++out.reparsed;
				if (pc.last_char != 0) { nx.begin = pc.stop; f.parse(pc.stop, pc.last_char, std::max(nx.end, pc.stop), nx); }
				else if (pc.stop >= nx.end) { nx.begin = pc.stop; nx.recs.clear(); nx.side.clear(); nx.stop = pc.stop; nx.last_char = 0; nx.stream_over = false; nx.end = std::max(nx.end, pc.stop); }
				else { nx.begin = pc.stop; f.parse(pc.stop, 0, nx.end, nx); }
			}
		}
	}
	{	// (pieces parsed again here copy their multi-line records too: the limit holds for what the stitched parse keeps in all)
		u64 side = 0;
		for (auto &pc : out.pieces) side += pc.side.size();
		if (side > side_limit) { out.pieces.clear(); out.too_wrapped = true; return; }
	}
	for (auto &pc : out.pieces) out.n_recs += pc.recs.size();
}
