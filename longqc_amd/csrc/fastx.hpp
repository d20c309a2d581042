// longqc_amd/csrc/fastx.hpp -- host-side FASTA/FASTQ (optionally gzip) record stream.
//
// Input semantics follow what the reference accepts (kseq.h:179-224 via bseq.c:56-102):
//   * a record starts at '>' or '@'; the name is the header up to the first whitespace;
//   * sequence lines are concatenated until a line that starts with '>', '@' or '+';
//   * one trailing '\r' per line is dropped; quality lines are read until they cover the sequence;
//   * a truncated quality string ends the stream.
// Reads go into one flat ReadBatch (bases, offsets, names) -- the layout the C ABI takes.
#pragma once
#include <zlib.h>
#include <string>
#include <vector>
#include <cstdint>
#include <cctype>
#include <stdexcept>

struct ReadBatch {
	std::vector<uint8_t> seq, qual;
	std::vector<uint64_t> seq_off{0};
	std::vector<char> names;
	std::vector<uint64_t> name_off{0};
	bool any_qual = false;
	uint32_t size() const { return (uint32_t)(seq_off.size() - 1); }
	uint64_t bases() const { return seq_off.back(); }
	void clear() { seq.clear(); qual.clear(); seq_off.assign(1, 0); names.clear(); name_off.assign(1, 0); any_qual = false; }
	const char *name(uint32_t i) const { return names.data() + name_off[i]; }
	void add(const std::string &nm, const std::string &s, const std::string &q, bool keep_qual)
	{
		seq.insert(seq.end(), s.begin(), s.end());
		if (keep_qual) {
			if (!q.empty()) { qual.insert(qual.end(), q.begin(), q.end()); any_qual = true; }
			else qual.insert(qual.end(), s.size(), 0);        // FASTA record: no qualities
		}
		seq_off.push_back(seq.size());
		names.insert(names.end(), nm.begin(), nm.end()); names.push_back('\0');
		name_off.push_back(names.size());
	}
};

class FastxReader {
	gzFile fp_ = nullptr;
	std::vector<unsigned char> buf_;
	int begin_ = 0, end_ = 0;
	bool eof_ = false;
	int last_char_ = 0;
	std::string name_, seq_, qual_;

	int getc_()
	{
		if (begin_ >= end_) {
			if (eof_) return -1;
			begin_ = 0;
			end_ = gzread(fp_, buf_.data(), (unsigned)buf_.size());
			if (end_ < (int)buf_.size()) eof_ = true;
			if (end_ <= 0) { end_ = 0; return -1; }
		}
		return buf_[begin_++];
	}
	// append up to (not including) the delimiter; space_delim: any whitespace, else '\n'.  Returns
	// false only if the stream was already exhausted; *dret = delimiter char or 0 at EOF.
	bool until_(bool space_delim, std::string &s, int *dret)
	{
		if (dret) *dret = 0;
		if (begin_ >= end_ && eof_) return false;
		bool got_any = false;                                  // a byte or a delimiter was consumed
		for (;;) {
			if (begin_ >= end_) {
				if (eof_) break;
				begin_ = 0;
				end_ = gzread(fp_, buf_.data(), (unsigned)buf_.size());
				if (end_ < (int)buf_.size()) eof_ = true;
				if (end_ <= 0) { end_ = 0; break; }
			}
			int i = begin_;
			if (space_delim) while (i < end_ && !isspace(buf_[i])) ++i;
			else while (i < end_ && buf_[i] != '\n') ++i;
			s.append((const char*)buf_.data() + begin_, (size_t)(i - begin_));
			if (i > begin_ || i < end_) got_any = true;
			begin_ = i + 1;
			if (i < end_) { if (dret) *dret = buf_[i]; break; }
		}
		if (!got_any) return false;                            // the stream ended exactly here: no record (kseq.h: ks_getuntil2 returns -1)
		if (!space_delim && s.size() > 1 && s.back() == '\r') s.pop_back();
		return true;
	}

public:
	explicit FastxReader(const std::string &path) : buf_(1 << 20)
	{
		fp_ = gzopen(path.c_str(), "r");
		if (!fp_) throw std::runtime_error("failed to open file '" + path + "'");
		gzbuffer(fp_, 1 << 20);
	}
	~FastxReader() { if (fp_) gzclose(fp_); }
	FastxReader(const FastxReader&) = delete;

	// true if a record was read into (name, seq, qual); qual is empty for FASTA
	bool next(std::string &name, std::string &seq, std::string &qual)
	{
		int c;
		if (last_char_ == 0) {
			while ((c = getc_()) != -1 && c != '>' && c != '@') {}
			if (c == -1) return false;
			last_char_ = c;
		}
		name.clear(); seq.clear(); qual.clear();
		if (!until_(true, name, &c)) return false;
		if (c != '\n') { std::string comment; until_(false, comment, nullptr); }
		while ((c = getc_()) != -1 && c != '>' && c != '+' && c != '@') {
			if (c == '\n') continue;
			seq.push_back((char)c);
			until_(false, seq, nullptr);
		}
		if (c == '>' || c == '@') last_char_ = c;
		if (c != '+') return true;
		while ((c = getc_()) != -1 && c != '\n') {}
		if (c == -1) return false;                      // no quality string: kseq returns -2
		while (until_(false, qual, nullptr) && qual.size() < seq.size()) {}
		last_char_ = 0;
		if (seq.size() != qual.size()) return false;    // truncated quality: kseq returns -2
		return true;
	}

	// mm_bseq_read2 (bseq.c:68-102): append records until their bases reach `chunk`; U -> T
	// (bseq.c:61-63).  Returns the number of records appended.
	// raw = true: plain kseq records, bases untouched (what `sdust` reads, sdust.c:199).
	uint32_t read_minibatch(int64_t chunk, ReadBatch &out, bool keep_qual, bool raw = false)
	{
		int64_t size = 0;
		uint32_t n = 0;
		while (next(name_, seq_, qual_)) {
			if (!raw) for (auto &ch : seq_) if (ch == 'u' || ch == 'U') --ch;
			out.add(name_, seq_, qual_, keep_qual);
			++n; size += (int64_t)seq_.size();
			if (size >= chunk) break;
		}
		return n;
	}
};
