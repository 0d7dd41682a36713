// lfLoader.h — loads a cols×rows grid of same-sized images named <row>_<col>.<ext> from a directory.
// Public interface source-compatible with the reference's LfLoader (reference src/lfLoader.h:7-41) with glm's vector
// types replaced by the PODs of vec.h.
#pragma once

#include <filesystem>
#include <set>
#include <string>
#include <vector>

#include "vec.h"

class LfLoader
{
    public:
        using DataGrid = std::vector<std::vector<std::vector<uint8_t>>>;
        lfi::IVec2 getColsRows() const
        {
            return colsRows;
        }
        void loadData(std::string path);
        size_t imageSize() const
        {
            return static_cast<size_t>(resolution.x) * resolution.y * resolution.z;
        }
        lfi::IVec3 imageResolution() const
        {
            return resolution;
        }
        size_t imageCount() const
        {
            return static_cast<size_t>(colsRows.x) * colsRows.y;
        }
        const std::vector<uint8_t> &image(lfi::IVec2 colRow) const
        {
            return grid[colRow.x][colRow.y];
        }

    private:
        lfi::IVec3 resolution{};
        lfi::IVec2 colsRows{};
        DataGrid grid;
        void initGrid(lfi::IVec2 inColsRows);
        const std::set<std::filesystem::path> listPath(std::string path) const;
        lfi::IVec2 parseFilename(std::string name) const;
        void loadImage(std::string path, lfi::IVec2 coords);
};
