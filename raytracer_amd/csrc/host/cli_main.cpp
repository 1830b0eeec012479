// `raytracer <scene.json5> <output image> [flags]`: the reference's command line (struct Opt,
// reference src/main.rs:32-71) over the C ABI of include/rt_mi355x.h.  The row loop of
// src/main.rs:117-154 becomes bands of rows rendered on the GPU from the top of the image down;
// the monitor thread (src/main.rs:191-246) keeps its job: progress bar on stderr and the .PART
// crash-recovery file, one record per finished row.
//
// Differences from the reference, all deliberate:
//  * --gpus / --devices render one image over several GPUs (rt_multi_*: samples split, one RCCL reduce per band).
//  * --seed / --precision / --device / --band-rows / --stats are new (the reference draws from
//    thread_rng, so its images are not reproducible; without --seed ours are not either).
//  * the monitor thread is always joined (the reference skips the join under --quiet and can exit
//    before the part file is removed).
//  * .png, .jpg/.jpeg, .ppm/.pnm, .pam, .bmp, .tga, .tiff/.tif and .ff outputs are encoded (image_out.cpp); .gif, .ico, .pgm, .pbm are not.
#include <sys/ioctl.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <optional>
#include <random>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "part_file.hpp"
#include "progress.hpp"
#include "rt_mi355x.h"

namespace rt {
void save_image(const std::string& path, uint32_t width, uint32_t height, const uint8_t* rgb);
}

namespace {

const char* kVersion = "raytracer 0.1.0 (rt_mi355x)";

const char* kHelp =
    "raytracer 0.1.0\n"
    "Raytracing in a weekend! (MI355X)\n"
    "\n"
    "USAGE:\n"
    "    raytracer [FLAGS] [OPTIONS] <input-file> <output-file>\n"
    "\n"
    "FLAGS:\n"
    "        --ascii-symbols-only    Do not use non-ASCII symbols\n"
    "    -h, --help                  Prints help information\n"
    "        --no-part-file          Don't save partial progress in a part file in case of a crash\n"
    "    -q, --quiet                 No informational messages printed to stderr\n"
    "        --recover-corrupt       Try to read as much of a corrupted part file as possible\n"
    "        --stats                 Print ray counts and GPU timings to stderr\n"
    "    -V, --version               Prints version information\n"
    "\n"
    "OPTIONS:\n"
    "        --band-rows <n>                 Rows per GPU batch (default: about 2^24 camera paths)\n"
    "        --device <n>                    GPU to render on [default: 0]\n"
    "        --devices <a,b,..>              GPUs to share the image (samples are split, one RCCL reduce per band)\n"
    "        --gpus <n>                      The same with GPUs 0 .. n-1\n"
    "    -m, --max-bounces <max-bounces>     Maximum number of bounces for any ray [default: 50]\n"
    "        --precision <f64|f32>           Arithmetic of the path tracer [default: f64]\n"
    "        --progress-bar-len <n>          Manually set length of progress bar\n"
    "        --progress-bar-samples <n>      Use this many of the most recent updates to determine time remaining\n"
    "                                        [default: 100]\n"
    "    -s, --ray-samples <ray-samples>     Rays per pixel [default: 100]\n"
    "    -r, --recover-from <recover-from>   Recover from part file\n"
    "        --seed <n>                      Seed of the counter-based generator (default: random)\n"
    "    -w, --width <width>                 Output image width [default: 600]\n"
    "\n"
    "ARGS:\n"
    "    <input-file>     Config file\n"
    "    <output-file>    Output file\n";

struct Opt {
    std::string input_file, output_file;
    uint32_t width = 600, ray_samples = 100, max_bounces = 50;
    bool quiet = false, ascii_symbols_only = false, recover_corrupt = false, no_part_file = false, stats = false;
    std::optional<std::string> recover_from;
    std::optional<size_t> progress_bar_len;
    uint32_t progress_bar_samples = 100;
    std::optional<uint64_t> seed;
    int precision = RT_PRECISION_F64;
    int device = 0;
    std::vector<int> devices;  // --gpus / --devices: more than one GPU (rt_multi_*)
    uint32_t band_rows = 0;
};

[[noreturn]] void usage_error(const std::string& msg) {
    std::fprintf(stderr,
                 "error: %s\n\nUSAGE:\n    raytracer [FLAGS] [OPTIONS] <input-file> <output-file>\n\n"
                 "For more information try --help\n",
                 msg.c_str());
    std::exit(1);
}

uint64_t parse_uint(const std::string& name, const std::string& s, uint64_t max) {
    if (s.empty()) usage_error("Invalid value for '" + name + "': cannot parse integer from empty string");
    uint64_t v = 0;
    for (char c : s) {
        if (c < '0' || c > '9') usage_error("Invalid value for '" + name + "': invalid digit found in string");
        uint64_t nv = v * 10 + (uint64_t)(c - '0');
        if (nv / 10 != v || nv > max) usage_error("Invalid value for '" + name + "': number too large to fit in target type");
        v = nv;
    }
    return v;
}

Opt parse_args(int argc, char** argv) {
    Opt o;
    std::vector<std::string> positional;
    bool only_positional = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (only_positional || a.size() < 2 || a[0] != '-') {
            positional.push_back(a);
            continue;
        }
        if (a == "--") {
            only_positional = true;
            continue;
        }
        std::string name;
        std::optional<std::string> inline_value;
        if (a[1] == '-') {
            size_t eq = a.find('=');
            name = a.substr(0, eq);
            if (eq != std::string::npos) inline_value = a.substr(eq + 1);
        } else {
            name = a.substr(0, 2);
            if (a.size() > 2) inline_value = a.substr(a[2] == '=' ? 3 : 2);
        }
        auto flag = [&](bool& dst) {
            if (inline_value && a[1] == '-') usage_error("The argument '" + name + "' doesn't take a value");
            dst = true;
            // combined short flags such as -qq are not used by the reference's option set
            if (inline_value && a[1] != '-') usage_error("Found argument '-" + *inline_value + "' which wasn't expected, or isn't valid in this context");
        };
        auto value = [&]() -> std::string {
            if (inline_value) return *inline_value;
            if (i + 1 >= argc) usage_error("The argument '" + name + " <value>' requires a value but none was supplied");
            return argv[++i];
        };
        if (name == "-h" || name == "--help") {
            std::fputs(kHelp, stdout);
            std::exit(0);
        } else if (name == "-V" || name == "--version") {
            std::printf("%s\n", kVersion);
            std::exit(0);
        } else if (name == "-w" || name == "--width") o.width = (uint32_t)parse_uint("--width <width>", value(), UINT32_MAX);
        else if (name == "-s" || name == "--ray-samples") o.ray_samples = (uint32_t)parse_uint("--ray-samples <ray-samples>", value(), UINT32_MAX);
        else if (name == "-m" || name == "--max-bounces") o.max_bounces = (uint32_t)parse_uint("--max-bounces <max-bounces>", value(), UINT32_MAX);
        else if (name == "-q" || name == "--quiet") flag(o.quiet);
        else if (name == "--ascii-symbols-only") flag(o.ascii_symbols_only);
        else if (name == "-r" || name == "--recover-from") o.recover_from = value();
        else if (name == "--recover-corrupt") flag(o.recover_corrupt);
        else if (name == "--progress-bar-len") o.progress_bar_len = (size_t)parse_uint("--progress-bar-len <progress-bar-len>", value(), UINT64_MAX);
        else if (name == "--progress-bar-samples") o.progress_bar_samples = (uint32_t)parse_uint("--progress-bar-samples <progress-bar-samples>", value(), UINT32_MAX);
        else if (name == "--no-part-file") flag(o.no_part_file);
        else if (name == "--stats") flag(o.stats);
        else if (name == "--seed") o.seed = parse_uint("--seed <n>", value(), UINT64_MAX);
        else if (name == "--device") o.device = (int)parse_uint("--device <n>", value(), 1023);
        else if (name == "--gpus") {
            const int n = (int)parse_uint("--gpus <n>", value(), 64);
            if (n < 1) usage_error("Invalid value for '--gpus <n>': at least 1");
            o.devices.clear();
            for (int g = 0; g < n; g++) o.devices.push_back(g);
        } else if (name == "--devices") {
            const std::string list = value();
            o.devices.clear();
            size_t at = 0;
            while (at <= list.size()) {
                const size_t comma = std::min(list.find(',', at), list.size());
                o.devices.push_back((int)parse_uint("--devices <a,b,..>", list.substr(at, comma - at), 1023));
                at = comma + 1;
            }
        } else if (name == "--band-rows") o.band_rows = (uint32_t)parse_uint("--band-rows <n>", value(), UINT32_MAX);
        else if (name == "--precision") {
            std::string v = value();
            if (v == "f64") o.precision = RT_PRECISION_F64;
            else if (v == "f32") o.precision = RT_PRECISION_F32;
            else usage_error("Invalid value for '--precision <f64|f32>': " + v);
        } else
            usage_error("Found argument '" + name + "' which wasn't expected, or isn't valid in this context");
    }
    if (positional.size() < 2) {
        std::string missing = positional.empty() ? "    <input-file>\n    <output-file>" : "    <output-file>";
        usage_error("The following required arguments were not provided:\n" + missing);
    }
    if (positional.size() > 2) usage_error("Found argument '" + positional[2] + "' which wasn't expected, or isn't valid in this context");
    o.input_file = positional[0];
    o.output_file = positional[1];
    return o;
}

// mpsc::channel<(u32, Vec<u8>)> of src/main.rs:90-91.
class RowChannel {
public:
    void send(uint32_t j, std::vector<uint8_t> row) {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.emplace_back(j, std::move(row));
        }
        cv_.notify_one();
    }
    void close() {
        {
            std::lock_guard<std::mutex> lk(m_);
            closed_ = true;
        }
        cv_.notify_all();
    }
    bool recv(std::pair<uint32_t, std::vector<uint8_t>>& out) {  // false: all senders gone
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !q_.empty() || closed_; });
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        return true;
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::pair<uint32_t, std::vector<uint8_t>>> q_;
    bool closed_ = false;
};

size_t terminal_width() {  // terminal_size(), src/main.rs:209-213
    for (int fd : {STDOUT_FILENO, STDERR_FILENO, STDIN_FILENO}) {
        struct winsize ws;
        if (isatty(fd) && ioctl(fd, TIOCGWINSZ, &ws) == 0 && ws.ws_col > 0) return ws.ws_col < 100 ? ws.ws_col : 100;
    }
    return 60;
}

// monitor_progress, src/main.rs:191-246.  Returns an error text, empty on success.
std::string monitor_progress(const Opt& opt, uint32_t width, uint32_t height, RowChannel& rows) {
    try {
        std::optional<rt::PartFile> part_file;
        if (!opt.no_part_file) {
            part_file.emplace(rt::PartFile::create(opt.output_file));
            part_file->write_header(width, height);
        }
        size_t bar_len = opt.progress_bar_len ? *opt.progress_bar_len : terminal_width();
        rt::TimedProgressBar progress(stderr, bar_len, "Rendering", opt.ascii_symbols_only ? " .:" : " -=\xE2\x89\xA1",
                                      opt.progress_bar_samples, height);
        bool complete = true;
        for (uint32_t n = 0; n < height; n++) {
            std::pair<uint32_t, std::vector<uint8_t>> msg;
            if (!rows.recv(msg)) {  // the renderer stopped early: keep the part file for --recover-from
                complete = false;
                break;
            }
            if (!opt.quiet) progress.update();
            if (part_file) {
                try {
                    part_file->write_part(msg.first, msg.second);
                } catch (const std::exception& e) {
                    std::fprintf(stderr, "\rError writing to part file: %s\nPart file may be corrupted\n", e.what());
                }
            }
        }
        if (part_file && complete) part_file->remove();
        if (!opt.quiet) progress.clear();
        return "";
    } catch (const std::exception& e) {
        return e.what();
    }
}

struct Fail : std::runtime_error {
    using std::runtime_error::runtime_error;
};

int run(const Opt& opt) {
    rt_scene* scene = nullptr;
    if (rt_scene_load_file(opt.input_file.c_str(), "", &scene) != RT_OK) throw Fail(rt_last_global_error());
    const uint32_t width = opt.width;
    const uint32_t height = rt_scene_image_height(scene, width);
    auto render_start = std::chrono::steady_clock::now();

    std::vector<std::optional<std::vector<uint8_t>>> base(height);
    if (opt.recover_from) base = rt::PartFile::read(*opt.recover_from, height, width, opt.recover_corrupt);

    rt_options options{};
    options.precision = opt.precision;
    rt_context* ctx = nullptr;
    rt_multi* multi = nullptr;  // several GPUs (--gpus / --devices); one GPU goes through rt_render_rows
    const bool many = opt.devices.size() > 1;
    if (many) {
        if (rt_multi_create(opt.devices.data(), (int)opt.devices.size(), &options, &multi) != RT_OK) throw Fail(rt_multi_last_error(nullptr));
    } else {
        if (rt_create(opt.devices.empty() ? opt.device : opt.devices[0], &options, &ctx) != RT_OK) throw Fail(rt_last_global_error());
    }
    struct CtxGuard {
        rt_context* c;
        rt_multi* m;
        rt_scene* s;
        ~CtxGuard() {
            rt_destroy(c);
            rt_multi_destroy(m);
            rt_scene_free(s);
        }
    } guard{ctx, multi, scene};
    if (many) {
        if (rt_multi_upload_scene(multi, rt_scene_get_desc(scene)) != RT_OK) throw Fail(rt_multi_last_error(multi));
    } else if (rt_upload_scene(ctx, rt_scene_get_desc(scene)) != RT_OK) {
        throw Fail(rt_last_error(ctx));
    }

    // Nothing has been written so far: a bad scene, part file or GPU leaves no new .PART behind.
    RowChannel channel;
    std::string monitor_error;
    std::thread monitor([&] { monitor_error = monitor_progress(opt, width, height, channel); });
    struct Joiner {
        RowChannel& c;
        std::thread& t;
        ~Joiner() {
            c.close();
            if (t.joinable()) t.join();
        }
    } joiner{channel, monitor};

    uint64_t seed;
    if (opt.seed) seed = *opt.seed;
    else {
        std::random_device rd;
        seed = ((uint64_t)rd() << 32) | rd();
    }
    uint32_t band = opt.band_rows;
    if (band == 0) {
        uint64_t per_row = (uint64_t)width * (opt.ray_samples ? opt.ray_samples : 1);
        band = (uint32_t)std::max<uint64_t>(1, (1ull << 24) / (per_row ? per_row : 1));
    }

    // Rows from the top of the picture (j = height-1) down, like the reversed row list of
    // src/main.rs:117-123; runs of rows missing from the part file are rendered in bands.
    std::vector<uint8_t> pixels((size_t)width * height * 3);
    const size_t row_bytes = (size_t)width * 3;
    auto place = [&](uint32_t j, const uint8_t* src) { std::memcpy(&pixels[(size_t)(height - 1 - j) * row_bytes], src, row_bytes); };
    rt_stats total{};
    std::vector<uint8_t> band_rgb;
    uint32_t hi = height;  // rows [hi, height) are done
    while (hi > 0) {
        if (base[hi - 1]) {
            place(hi - 1, base[hi - 1]->data());
            channel.send(hi - 1, std::move(*base[hi - 1]));
            hi--;
            continue;
        }
        uint32_t lo = hi - 1;
        // one GPU: a band per call; several: the whole run of missing rows per call, cut into bands inside, so that a
        // band's reduce runs under the next band's tracing (rows arrive through the call-back, band by band)
        while (lo > 0 && !base[lo - 1] && (many || hi - (lo - 1) <= band)) lo--;
        band_rgb.resize((size_t)(hi - lo) * row_bytes);
        rt_stats st{};
        if (many) {
            // The library's bands run upwards from `lo`; the reference hands rows over from the top down.  The bar and
            // the part file only need every row once (records carry their row number), so they go in band order.
            struct Sink {
                RowChannel* channel;
                const std::vector<uint8_t>* rgb;
                uint32_t lo;
                size_t row_bytes;
                decltype(place)* put;
            } sink{&channel, &band_rgb, lo, row_bytes, &place};
            auto on_band = [](void* user, uint32_t b0, uint32_t b1) {
                Sink& k = *(Sink*)user;
                for (uint32_t j = b1; j-- > b0;) {
                    const uint8_t* row = &(*k.rgb)[(size_t)(j - k.lo) * k.row_bytes];
                    (*k.put)(j, row);
                    k.channel->send(j, std::vector<uint8_t>(row, row + k.row_bytes));
                }
            };
            if (rt_multi_render_rows_cb(multi, width, height, lo, hi, opt.ray_samples, opt.max_bounces, seed, band, band_rgb.data(), nullptr,
                                        &st, on_band, &sink) != RT_OK)
                throw Fail(rt_multi_last_error(multi));
        } else if (rt_render_rows(ctx, width, height, lo, hi, opt.ray_samples, opt.max_bounces, seed, band_rgb.data(), nullptr, &st) != RT_OK) {
            throw Fail(rt_last_error(ctx));
        }
        total.rays += st.rays;
        total.paths += st.paths;
        total.ms_total += st.ms_total;
        total.ms_extend += st.ms_extend;
        total.ms_shade += st.ms_shade;
        total.ms_tail += st.ms_tail;
        total.batches += st.batches;
        for (uint32_t j = hi; !many && j-- > lo;) {
            const uint8_t* row = &band_rgb[(size_t)(j - lo) * row_bytes];
            place(j, row);
            channel.send(j, std::vector<uint8_t>(row, row + row_bytes));
        }
        hi = lo;
    }
    channel.close();
    monitor.join();
    if (!monitor_error.empty()) throw Fail(monitor_error);

    rt::save_image(opt.output_file, width, height, pixels.data());

    if (opt.stats)
        std::fprintf(stderr, "seed %llu: %llu paths, %llu rays, %.2f ms on the GPU (extend %.2f, shade %.2f, tail %.2f), %.1f Mrays/s\n",
                     (unsigned long long)seed, (unsigned long long)total.paths, (unsigned long long)total.rays, total.ms_total,
                     total.ms_extend, total.ms_shade, total.ms_tail, total.ms_total > 0 ? (double)total.rays / total.ms_total * 1e-3 : 0.0);
    if (!opt.quiet) {
        auto elapsed = std::chrono::duration_cast<std::chrono::seconds>(std::chrono::steady_clock::now() - render_start).count();
        std::fprintf(stderr, "Completed in %lld:%02lld\n", (long long)(elapsed / 60), (long long)(elapsed % 60));
    }
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    Opt opt = parse_args(argc, argv);
    try {
        return run(opt);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "Error: %s\n", e.what());  // anyhow's report of main's Err
        return 1;
    }
}
