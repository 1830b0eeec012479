// C ABI over the host-side file formats of the command line (include/rt_mi355x.h, "host I/O"):
// .PART recovery files, output images and the progress bar.  No GPU work happens here.
#include <unistd.h>

#include <cerrno>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../../include/rt_mi355x.h"
#include "part_file.hpp"
#include "progress.hpp"

namespace rt {
void set_global_error(const std::string& msg);
void save_image(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgb);
}  // namespace rt

struct rt_part_file {
    rt::PartFile file;
};

struct rt_progress {
    std::FILE* stream;
    std::unique_ptr<rt::TimedProgressBar> bar;
};

namespace {
template <typename F>
int guarded(int code_on_error, F&& f) {
    try {
        f();
        return RT_OK;
    } catch (const std::exception& e) {
        rt::set_global_error(e.what());
        return code_on_error;
    }
}
}  // namespace

extern "C" {

int rt_part_file_create(const char* image_name, uint32_t width, uint32_t height, rt_part_file** out) {
    if (!image_name || !out) {
        rt::set_global_error("rt_part_file_create: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    return guarded(RT_ERR_IO, [&] {
        std::unique_ptr<rt_part_file> p(new rt_part_file{rt::PartFile::create(image_name)});
        p->file.write_header(width, height);
        *out = p.release();
    });
}

const char* rt_part_file_path(const rt_part_file* part) { return part ? part->file.path().c_str() : ""; }

int rt_part_file_write(rt_part_file* part, uint32_t line_number, const uint8_t* row, size_t row_bytes) {
    if (!part || (!row && row_bytes)) {
        rt::set_global_error("rt_part_file_write: null argument");
        return RT_ERR_INVALID;
    }
    return guarded(RT_ERR_IO, [&] { part->file.write_part(line_number, std::vector<uint8_t>(row, row + row_bytes)); });
}

int rt_part_file_finish(rt_part_file* part, int remove) {
    if (!part) return RT_OK;
    std::unique_ptr<rt_part_file> owner(part);
    if (!remove) return RT_OK;
    return guarded(RT_ERR_IO, [&] { owner->file.remove(); });
}

int rt_part_file_read(const char* name, uint32_t image_height, uint32_t image_width, int recover_corrupt,
                      uint8_t* rows, uint8_t* present) {
    if (!name || !rows || !present) {
        rt::set_global_error("rt_part_file_read: null argument");
        return RT_ERR_INVALID;
    }
    return guarded(RT_ERR_IO, [&] {
        auto list = rt::PartFile::read(name, image_height, image_width, recover_corrupt != 0);
        const size_t row_bytes = (size_t)image_width * 3;
        for (uint32_t j = 0; j < image_height; j++) {
            present[j] = list[j] ? 1 : 0;
            if (list[j]) std::memcpy(rows + (size_t)j * row_bytes, list[j]->data(), row_bytes);
        }
    });
}

int rt_save_image(const char* path, uint32_t width, uint32_t height, const uint8_t* rgb8) {
    if (!path || !rgb8) {
        rt::set_global_error("rt_save_image: null argument");
        return RT_ERR_INVALID;
    }
    return guarded(RT_ERR_IO, [&] { rt::save_image(path, width, height, rgb8); });
}

int rt_progress_create(int fd, size_t length, const char* label, const char* glyphs, uint32_t number_samples,
                       uint32_t times_will_be_updated, rt_progress** out) {
    if (!label || !glyphs || !out) {
        rt::set_global_error("rt_progress_create: null argument");
        return RT_ERR_INVALID;
    }
    *out = nullptr;
    return guarded(RT_ERR_INVALID, [&] {
        int own = ::dup(fd);
        std::FILE* stream = own >= 0 ? ::fdopen(own, "w") : nullptr;
        if (!stream) {
            if (own >= 0) ::close(own);
            throw std::runtime_error(std::string("rt_progress_create: ") + std::strerror(errno));
        }
        std::unique_ptr<rt_progress> p(new rt_progress{stream, nullptr});
        try {
            p->bar.reset(new rt::TimedProgressBar(stream, length, label, glyphs, number_samples, times_will_be_updated));
        } catch (...) {
            std::fclose(stream);
            throw;
        }
        *out = p.release();
    });
}

void rt_progress_update(rt_progress* p) {
    if (p) p->bar->update();
}

void rt_progress_clear(rt_progress* p) {
    if (p) p->bar->clear();
}

void rt_progress_free(rt_progress* p) {
    if (!p) return;
    std::fclose(p->stream);
    delete p;
}

}  // extern "C"
