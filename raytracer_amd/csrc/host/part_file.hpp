// .PART crash-recovery files, byte-compatible with reference src/part_file.rs:
//   header  "<W> <H>\n"                                   (written by monitor_progress, src/main.rs:197-200)
//   record  "L<j> " + 3*W raw RGB bytes + "\n", flushed   (write_part, src/part_file.rs:32-39)
// j is the bottom-up row index of the row loop (src/main.rs:121-134), not the output image row.
#pragma once
#include <cstdint>
#include <cstdio>
#include <optional>
#include <string>
#include <vector>

namespace rt {

class PartFile {
public:
    // PartFile::new (src/part_file.rs:13-30): "<basename(image)>.<n>.PART" in the CWD, first free n.
    static PartFile create(const std::string& image_name);
    PartFile(PartFile&& o) noexcept;
    PartFile(const PartFile&) = delete;
    ~PartFile();
    void write_header(uint32_t width, uint32_t height);
    void write_part(uint32_t line_number, const std::vector<uint8_t>& part);  // src/part_file.rs:32-39
    void remove();  // src/main.rs:239-241
    const std::string& path() const { return path_; }

    // PartFile::read (src/part_file.rs:41-75): rows present in the file; throws std::runtime_error
    // with the reference's messages.  recover_corrupt: stop silently at the first bad record.
    static std::vector<std::optional<std::vector<uint8_t>>> read(const std::string& name, uint32_t image_height,
                                                                 uint32_t image_width, bool recover_corrupt);

private:
    PartFile(std::FILE* f, std::string path) : file_(f), path_(std::move(path)) {}
    std::FILE* file_;
    std::string path_;
};

}  // namespace rt
