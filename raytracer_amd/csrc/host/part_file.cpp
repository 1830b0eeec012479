// See part_file.hpp.  Follows reference src/part_file.rs line by line in behaviour.
#include "part_file.hpp"

#include <sys/stat.h>

#include <cerrno>
#include <cstring>
#include <stdexcept>

namespace rt {

namespace {

bool exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}

std::string basename_of(const std::string& p) {  // Path::file_name
    size_t end = p.find_last_not_of('/');
    if (end == std::string::npos) throw std::runtime_error("bad image name");
    size_t start = p.find_last_of('/', end);
    std::string base = p.substr(start == std::string::npos ? 0 : start + 1, end - (start == std::string::npos ? 0 : start + 1) + 1);
    if (base.empty() || base == "." || base == "..") throw std::runtime_error("bad image name");
    return base;
}

struct Eof {};

uint8_t get_byte(std::FILE* f) {  // file_get_byte, src/part_file.rs:101-105
    int c = std::fgetc(f);
    if (c == EOF) {
        if (std::ferror(f)) throw std::runtime_error(std::string("Read error ") + std::strerror(errno));
        throw Eof{};
    }
    return (uint8_t)c;
}

size_t read_num(std::FILE* f) {  // file_read_num, src/part_file.rs:107-118
    std::string num;
    for (;;) {
        uint8_t b;
        try {
            b = get_byte(f);
        } catch (const Eof&) {
            throw std::runtime_error("failed to fill whole buffer");
        }
        if (b == ' ' || b == '\n') break;
        num += (char)b;
    }
    if (num.empty()) throw std::runtime_error("cannot parse integer from empty string");
    size_t v = 0;
    for (char c : num) {
        if (c < '0' || c > '9') throw std::runtime_error("invalid digit found in string");
        size_t nv = v * 10 + (size_t)(c - '0');
        if (nv < v) throw std::runtime_error("number too large to fit in target type");
        v = nv;
    }
    return v;
}

}  // namespace

PartFile PartFile::create(const std::string& image_name) {
    std::string base = basename_of(image_name);
    std::string path;
    for (unsigned i = 0;; i++) {
        path = base + "." + std::to_string(i) + ".PART";
        if (!exists(path)) break;
    }
    std::FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot create " + path + ": " + std::strerror(errno));
    return PartFile(f, path);
}

PartFile::PartFile(PartFile&& o) noexcept : file_(o.file_), path_(std::move(o.path_)) { o.file_ = nullptr; }

PartFile::~PartFile() {
    if (file_) std::fclose(file_);
}

void PartFile::write_header(uint32_t width, uint32_t height) {
    std::string h = std::to_string(width) + " " + std::to_string(height) + "\n";
    if (std::fwrite(h.data(), 1, h.size(), file_) != h.size()) throw std::runtime_error("write error on " + path_);
}

void PartFile::write_part(uint32_t line_number, const std::vector<uint8_t>& part) {
    std::string head = "L" + std::to_string(line_number) + " ";
    if (std::fwrite(head.data(), 1, head.size(), file_) != head.size() ||
        std::fwrite(part.data(), 1, part.size(), file_) != part.size() || std::fputc('\n', file_) == EOF ||
        std::fflush(file_) != 0)
        throw std::runtime_error(std::string("write error on ") + path_ + ": " + std::strerror(errno));
}

void PartFile::remove() {
    if (file_) {
        std::fclose(file_);
        file_ = nullptr;
    }
    if (std::remove(path_.c_str()) != 0) throw std::runtime_error("cannot remove " + path_ + ": " + std::strerror(errno));
}

std::vector<std::optional<std::vector<uint8_t>>> PartFile::read(const std::string& name, uint32_t image_height,
                                                                uint32_t image_width, bool recover_corrupt) {
    std::FILE* f = std::fopen(name.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + name + ": " + std::strerror(errno));
    struct Closer {
        std::FILE* f;
        ~Closer() { std::fclose(f); }
    } closer{f};
    std::vector<std::optional<std::vector<uint8_t>>> list(image_height);
    size_t width2, height2;
    try {
        width2 = read_num(f);
    } catch (const std::exception& e) {
        throw std::runtime_error(std::string("Reading width: ") + e.what());
    }
    try {
        height2 = read_num(f);
    } catch (const std::exception& e) {
        throw std::runtime_error(std::string("Reading height: ") + e.what());
    }
    if (image_height != height2 || image_width != width2)
        throw std::runtime_error("Image dimensions expected to be " + std::to_string(image_width) + "x" +
                                 std::to_string(image_height) + ", but recovery file says they're " +
                                 std::to_string(width2) + "x" + std::to_string(height2));
    for (;;) {  // read_part, src/part_file.rs:77-99
        try {
            uint8_t b;
            try {
                b = get_byte(f);
            } catch (const Eof&) {
                break;  // clean end of file
            }
            if (b != 'L') throw std::runtime_error("Missing leading L");
            size_t line_number = read_num(f);
            std::vector<uint8_t> buf((size_t)image_width * 3);
            if (std::fread(buf.data(), 1, buf.size(), f) != buf.size()) throw std::runtime_error("failed to fill whole buffer");
            // The reference indexes list[line_number] unchecked (a panic on a bad index): reported as an error here.
            if (line_number >= list.size()) throw std::runtime_error("line number " + std::to_string(line_number) + " out of range");
            list[line_number] = std::move(buf);
            uint8_t nl;
            try {
                nl = get_byte(f);
            } catch (const Eof&) {
                throw std::runtime_error("failed to fill whole buffer");
            }
            if (nl != '\n') throw std::runtime_error("Missing newline on line " + std::to_string(line_number));
        } catch (const std::runtime_error&) {
            if (recover_corrupt) break;
            throw;
        }
    }
    return list;
}

}  // namespace rt
