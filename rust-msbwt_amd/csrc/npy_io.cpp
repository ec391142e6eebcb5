#include "npy_io.hpp"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <memory>

namespace msbwt {
namespace {

struct FileCloser {
    void operator()(FILE *f) const { if (f) std::fclose(f); }
};
using File = std::unique_ptr<FILE, FileCloser>;

// The reference turns the python-dict header into JSON by textual substitution
// (rle_bwt.rs:115-122) and asks serde_json for ["shape"][0].as_u64(); whatever serde_json
// rejects is a panic.  This is a strict reader for the same grammar after the same
// substitutions: object / array / string / number / true / false / null.
class HeaderReader {
  public:
    explicit HeaderReader(std::string text) : s_(std::move(text)) {}

    bool shape0(uint64_t *out) {
        pos_ = 0;
        found_ = false;
        if (!value(0, false, -1)) return false;
        ws();
        if (pos_ != s_.size() || !found_) return false;
        *out = shape0_;
        return true;
    }

  private:
    void ws() {
        while (pos_ < s_.size() && (s_[pos_] == ' ' || s_[pos_] == '\n' || s_[pos_] == '\t' || s_[pos_] == '\r')) ++pos_;
    }
    bool lit(const char *word) {
        size_t n = std::strlen(word);
        if (s_.compare(pos_, n, word) != 0) return false;
        pos_ += n;
        return true;
    }
    bool str(std::string *out) {
        if (pos_ >= s_.size() || s_[pos_] != '"') return false;
        ++pos_;
        out->clear();
        while (pos_ < s_.size() && s_[pos_] != '"') {
            if (s_[pos_] == '\\') {
                if (++pos_ >= s_.size()) return false;
            }
            out->push_back(s_[pos_++]);
        }
        if (pos_ >= s_.size()) return false;
        ++pos_;
        return true;
    }
    bool number(bool *is_u64, uint64_t *v) {
        bool neg = false, integral = true;
        if (s_[pos_] == '-') { neg = true; ++pos_; }
        if (pos_ >= s_.size() || !std::isdigit(static_cast<unsigned char>(s_[pos_]))) return false;
        *v = 0;
        while (pos_ < s_.size() && std::isdigit(static_cast<unsigned char>(s_[pos_]))) *v = *v * 10 + uint64_t(s_[pos_++] - '0');
        while (pos_ < s_.size() && (s_[pos_] == '.' || s_[pos_] == 'e' || s_[pos_] == 'E' || s_[pos_] == '+' ||
                                    s_[pos_] == '-' || std::isdigit(static_cast<unsigned char>(s_[pos_])))) {
            integral = false;
            ++pos_;
        }
        *is_u64 = integral && !neg;
        return true;
    }
    // depth 0 = the top-level value; in_shape = this value belongs to top-level key "shape";
    // elem = index inside the shape array, or -1
    bool value(int depth, bool in_shape, int elem) {
        ws();
        if (pos_ >= s_.size()) return false;
        const char c = s_[pos_];
        if (c == '{') {
            ++pos_;
            ws();
            if (pos_ < s_.size() && s_[pos_] == '}') { ++pos_; return true; }
            for (;;) {
                std::string key;
                ws();
                if (!str(&key)) return false;
                ws();
                if (pos_ >= s_.size() || s_[pos_] != ':') return false;
                ++pos_;
                if (!value(depth + 1, depth == 0 && key == "shape", -1)) return false;
                ws();
                if (pos_ < s_.size() && s_[pos_] == ',') { ++pos_; continue; }
                if (pos_ < s_.size() && s_[pos_] == '}') { ++pos_; return true; }
                return false;
            }
        }
        if (c == '[') {
            ++pos_;
            ws();
            if (pos_ < s_.size() && s_[pos_] == ']') { ++pos_; return true; }
            for (int idx = 0;; ++idx) {
                if (!value(depth + 1, false, (in_shape && depth == 1) ? idx : -1)) return false;
                ws();
                if (pos_ < s_.size() && s_[pos_] == ',') { ++pos_; continue; }
                if (pos_ < s_.size() && s_[pos_] == ']') { ++pos_; return true; }
                return false;
            }
        }
        if (c == '"') {
            std::string tmp;
            return str(&tmp);
        }
        if (c == '-' || std::isdigit(static_cast<unsigned char>(c))) {
            bool is_u64;
            uint64_t v;
            if (!number(&is_u64, &v)) return false;
            if (elem == 0 && is_u64) { found_ = true; shape0_ = v; }
            return true;
        }
        return lit("true") || lit("false") || lit("null");
    }

    std::string s_;
    size_t pos_ = 0;
    bool found_ = false;
    uint64_t shape0_ = 0;
};

void substitute(std::string *s, const char *from, const char *to) {
    const size_t lf = std::strlen(from), lt = std::strlen(to);
    for (size_t p = s->find(from); p != std::string::npos; p = s->find(from, p + lt)) s->replace(p, lf, to);
}

bool valid_utf8(const std::string &s) {
    size_t i = 0;
    while (i < s.size()) {
        const unsigned char c = static_cast<unsigned char>(s[i]);
        int extra = c < 0x80 ? 0 : (c >> 5) == 0x6 ? 1 : (c >> 4) == 0xE ? 2 : (c >> 3) == 0x1E ? 3 : -1;
        if (extra < 0 || i + size_t(extra) >= s.size() + (extra == 0 ? 1 : 0)) return false;
        for (int k = 1; k <= extra; ++k)
            if ((static_cast<unsigned char>(s[i + size_t(k)]) >> 6) != 2) return false;
        i += size_t(extra) + 1;
    }
    return true;
}

}  // namespace

namespace {

// Opens `path`, validates the header exactly as the reference does (rle_bwt.rs:84-136) and
// reports where the payload starts and how long it is.  The FILE is left positioned there.
NpyStatus open_npy(const std::string &path, File *file, uint64_t *data_offset, uint64_t *payload_len, std::string *msg) {
    struct stat st;
    if (::stat(path.c_str(), &st) != 0) {
        *msg = "cannot stat " + path + ": " + std::strerror(errno);
        return NpyStatus::kIo;
    }
    const uint64_t file_size = uint64_t(st.st_size);
    File f(std::fopen(path.c_str(), "rb"));
    if (!f) {
        *msg = "cannot open " + path + ": " + std::strerror(errno);
        return NpyStatus::kIo;
    }
    unsigned char fixed[10];
    if (std::fread(fixed, 1, 10, f.get()) != 10) {
        *msg = "could not read initial 10 bytes of header for file " + path;
        return NpyStatus::kBadHeader;
    }
    // magic, version and dtype are deliberately not checked: the reference does not either
    const size_t header_len = size_t(fixed[8]) + 256 * size_t(fixed[9]);
    const size_t offset = (10 + header_len + 15) / 16 * 16;
    std::string header(offset - 10, '\0');
    if (std::fread(&header[0], 1, header.size(), f.get()) != header.size()) {
        *msg = "could not read bytes 10-" + std::to_string(offset) + " of header for file " + path;
        return NpyStatus::kUnexpectedEof;
    }
    if (!valid_utf8(header)) {
        *msg = "header of " + path + " is not UTF-8";
        return NpyStatus::kBadHeader;
    }
    substitute(&header, "'", "\"");
    substitute(&header, "False", "false");
    substitute(&header, "(", "[");
    substitute(&header, ")", "]");
    substitute(&header, ", }", "}");
    substitute(&header, ", ]", "]");
    substitute(&header, ",]", "]");
    uint64_t expected = 0;
    if (!HeaderReader(header).shape0(&expected)) {
        *msg = "error while parsing header string: " + header;
        return NpyStatus::kBadHeader;
    }
    const uint64_t on_disk = file_size - offset;
    if (expected != on_disk) {
        *msg = "header indicates shape of " + std::to_string(expected) + ", but remaining file size is " + std::to_string(on_disk);
        return NpyStatus::kUnexpectedEof;
    }
    *data_offset = offset;
    *payload_len = on_disk;
    *file = std::move(f);
    return NpyStatus::kOk;
}

}  // namespace

NpyStatus read_npy_payload(const std::string &path, std::vector<uint8_t> *payload, std::string *msg) {
    File f;
    uint64_t offset = 0, on_disk = 0;
    const NpyStatus st = open_npy(path, &f, &offset, &on_disk, msg);
    if (st != NpyStatus::kOk) return st;
    payload->resize(on_disk);
    const size_t got = on_disk ? std::fread(payload->data(), 1, on_disk, f.get()) : 0;
    if (got != on_disk) {
        *msg = "only read " + std::to_string(got) + " of " + std::to_string(on_disk) + " bytes of BWT body for file " + path;
        return NpyStatus::kUnexpectedEof;
    }
    return NpyStatus::kOk;
}

MappedPayload::~MappedPayload() {
    if (base_) ::munmap(base_, map_len_);
}

NpyStatus map_npy_payload(const std::string &path, MappedPayload *out, std::string *msg) {
    File f;
    uint64_t offset = 0, on_disk = 0;
    const NpyStatus st = open_npy(path, &f, &offset, &on_disk, msg);
    if (st != NpyStatus::kOk) return st;
    if (on_disk == 0) return NpyStatus::kOk;  // empty BWT: nothing to map
    const size_t len = size_t(offset + on_disk);
    void *base = ::mmap(nullptr, len, PROT_READ, MAP_PRIVATE, ::fileno(f.get()), 0);
    if (base == MAP_FAILED) {
        *msg = "cannot map " + path + ": " + std::strerror(errno);
        return NpyStatus::kIo;
    }
    ::madvise(base, len, MADV_SEQUENTIAL);
    out->base_ = base;
    out->map_len_ = len;
    out->payload_ = static_cast<const uint8_t *>(base) + offset;
    out->size_ = size_t(on_disk);
    return NpyStatus::kOk;
}

NpyStatus write_npy_payload(const std::string &path, const uint8_t *payload, size_t n, std::string *msg) {
    // 96 bytes: magic, v1.0, header_len 0x56, the dict text, space padding, '\n'
    std::string head("\x93NUMPY\x01\x00\x56\x00", 10);
    head += "{'descr': '|u1', 'fortran_order': False, 'shape': (" + std::to_string(n) + ", ), }";
    if (head.size() > 95) {
        *msg = "payload length does not fit the fixed 96-byte header";
        return NpyStatus::kBadHeader;
    }
    head.resize(95, ' ');
    head.push_back('\n');
    File f(std::fopen(path.c_str(), "wb"));
    if (!f) {
        *msg = "cannot create " + path + ": " + std::strerror(errno);
        return NpyStatus::kIo;
    }
    bool ok = std::fwrite(head.data(), 1, head.size(), f.get()) == head.size();
    ok = ok && (n == 0 || std::fwrite(payload, 1, n, f.get()) == n);
    ok = (std::fclose(f.release()) == 0) && ok;
    if (!ok) {
        *msg = "short write to " + path;
        return NpyStatus::kIo;
    }
    return NpyStatus::kOk;
}

}  // namespace msbwt
