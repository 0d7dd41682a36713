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

const std::set<std::filesystem::path> LfLoader::listPath(std::string path) const
{
    if(!std::filesystem::exists(path))
        throw std::runtime_error("The path " + path + " does not exist!");
    if(!std::filesystem::is_directory(path))
        throw std::runtime_error("The path " + path + " does not lead to a directory!");

    std::set<std::filesystem::path> sorted;
    for(const auto &file : std::filesystem::directory_iterator(path))
        sorted.insert(file.path().filename());
    return sorted;
}

// "<row>_<col>.<ext>": the first number is the vertical index, the second the horizontal one
// (reference src/lfLoader.cpp:22-31; its help text says column_row but the code does this)
lfi::IVec2 LfLoader::parseFilename(std::string name) const
{
    auto delimiterPos = name.find('_');
    if(delimiterPos == std::string::npos)
        throw std::runtime_error("File " + name + " is not named properly as column_row.extension!");
    auto extensionPos = name.find('.');
    auto row = name.substr(0, delimiterPos);
    auto col = name.substr(delimiterPos + 1, extensionPos == std::string::npos ? std::string::npos : extensionPos - delimiterPos - 1);
    try
    {
        return {std::stoi(row), std::stoi(col)};
    }
    catch(const std::exception &)
    {
        throw std::runtime_error("File " + name + " is not named properly as column_row.extension!");
    }
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
