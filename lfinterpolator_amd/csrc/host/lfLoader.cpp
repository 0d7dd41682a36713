// lfLoader.cpp — see lfLoader.h.  Behaviour follows reference src/lfLoader.cpp:8-67 with its defect D5 fixed: the grid
// size is the maximum over ALL file names (the reference takes it from the lexicographically last name, transposed), so
// non-square grids and unpadded numbers work; every image must have the same resolution and every grid cell a file.
#include "lfLoader.h"

#include <algorithm>
#include <atomic>
#include <iostream>
#include <mutex>
#include <stdexcept>
#include <thread>

#include "image_io.h"
#include "loadingbar.hpp"

// The file names of a directory, ordered.  Error texts as the reference prints them (src/lfLoader.cpp:8-20).
const std::set<std::filesystem::path> LfLoader::listPath(std::string path) const
{
    std::error_code ec;
    const std::filesystem::file_status st = std::filesystem::status(path, ec);
    if(ec || st.type() == std::filesystem::file_type::not_found)
        throw std::runtime_error("The path " + path + " does not exist!");
    if(st.type() != std::filesystem::file_type::directory)
        throw std::runtime_error("The path " + path + " does not lead to a directory!");
    std::set<std::filesystem::path> names;
    for(std::filesystem::directory_iterator it(path, ec), end; !ec && it != end; it.increment(ec))
        names.insert(it->path().filename());
    if(ec)
        throw std::runtime_error("The path " + path + " cannot be listed: " + ec.message());
    return names;
}

// "<row>_<col>.<ext>": two decimal numbers around the first underscore, the first is the vertical index, the second the horizontal one
// (what reference src/lfLoader.cpp:22-31 does; its help text says column_row).  Anything else — no underscore, no digits, trailing
// characters before the extension — is refused with the reference's message.
lfi::IVec2 LfLoader::parseFilename(std::string name) const
{
    const auto refuse = [&]() -> lfi::IVec2 { throw std::runtime_error("File " + name + " is not named properly as column_row.extension!"); };
    const auto number = [&](size_t &pos, int &out) {
        const size_t start = pos;
        long v = 0;
        while(pos < name.size() && name[pos] >= '0' && name[pos] <= '9' && v < 100000000)
            v = v * 10 + (name[pos++] - '0');
        out = static_cast<int>(v);
        return pos > start && v < 100000000;
    };
    size_t pos = 0;
    int row = 0, col = 0;
    if(!number(pos, row) || pos >= name.size() || name[pos] != '_')
        return refuse();
    pos++;
    if(!number(pos, col) || (pos < name.size() && name[pos] != '.'))
        return refuse();
    return {row, col};
}

void LfLoader::loadImage(std::string path, lfi::IVec2 coords)
{
    constexpr int RGBA_CHANNELS{4};
    lfi::Image img = lfi::loadImage(path);
    if(resolution.x != 0 && (resolution.x != img.width || resolution.y != img.height))
        throw std::runtime_error("Image " + path + " does not have the same resolution as the others");
    resolution = {img.width, img.height, RGBA_CHANNELS};
    grid[coords.x][coords.y] = std::move(img.pixels);
}

void LfLoader::initGrid(lfi::IVec2 inColsRows)
{
    colsRows = inColsRows;
    grid.assign(colsRows.x, std::vector<std::vector<uint8_t>>(colsRows.y));
}

void LfLoader::loadData(std::string path)
{
    auto files = listPath(path);
    if(files.empty())
        throw std::runtime_error("The input directory is empty!");
    lfi::IVec2 extent{0, 0};
    for(auto const &file : files)
    {
        auto rowCol = parseFilename(file.string());
        if(rowCol.x < 0 || rowCol.y < 0)
            throw std::runtime_error("File " + file.string() + " is not named properly as column_row.extension!");
        extent.x = std::max(extent.x, rowCol.y + 1); // cols from the second number
        extent.y = std::max(extent.y, rowCol.x + 1); // rows from the first
    }
    resolution = {};
    initGrid(extent);

    std::cout << "Loading images..." << std::endl;
    LoadingBar bar(files.size());
    // decode on several threads (the reference decodes one file after another with stb_image); every image goes to its own
    // grid cell, the shared resolution is checked afterwards
    std::vector<std::filesystem::path> list(files.begin(), files.end());
    std::vector<lfi::Image> decoded(list.size());
    std::vector<std::string> errors(list.size());
    const unsigned workers = std::max(1u, std::min<unsigned>(16u, std::thread::hardware_concurrency()));
    std::atomic<size_t> next{0};
    std::mutex barMutex;
    std::vector<std::thread> pool;
    for(unsigned w = 0; w < std::min<size_t>(workers, list.size()); w++)
        pool.emplace_back([&] {
            for(size_t i = next++; i < list.size(); i = next++)
            {
                try
                {
                    decoded[i] = lfi::loadImage((std::filesystem::path(path) / list[i]).string());
                }
                catch(const std::exception &e)
                {
                    errors[i] = e.what();
                }
                std::lock_guard<std::mutex> lock(barMutex);
                bar.add();
            }
        });
    for(auto &t : pool)
        t.join();
    constexpr int RGBA_CHANNELS{4};
    for(size_t i = 0; i < list.size(); i++)
    {
        if(!errors[i].empty())
            throw std::runtime_error(errors[i]);
        auto rowCol = parseFilename(list[i].string());
        if(resolution.x != 0 && (resolution.x != decoded[i].width || resolution.y != decoded[i].height))
            throw std::runtime_error("Image " + (std::filesystem::path(path) / list[i]).string() + " does not have the same resolution as the others");
        resolution = {decoded[i].width, decoded[i].height, RGBA_CHANNELS};
        grid[rowCol.y][rowCol.x] = std::move(decoded[i].pixels);
    }
    for(int col = 0; col < colsRows.x; col++)
        for(int row = 0; row < colsRows.y; row++)
            if(grid[col][row].empty())
                throw std::runtime_error("The grid image " + std::to_string(row) + "_" + std::to_string(col) + " is missing!");
}
