// ref_shim.cpp — driver around the parts of the REFERENCE that compile from their own sources in this image:
//   src/topn.h (TopN), src/defs.{h,cpp} (type conversions, ExtractPointsInBucket, ConvertTopKToNNVec, Timer),
//   src/metis_io.cpp (clusters / partition text formats).
// TEST INFRASTRUCTURE ONLY.  Built by oracle/Makefile (`make ref`) into oracle/_ref/ref_core, compiling the reference
// sources WHERE THEY LIE under $(REFERENCE)/src; nothing from the reference is copied into this repository and no
// stand-in headers are written.  defs.cpp's ClosestLeaders references distance() from src/dist.cpp, which cannot be
// built (it includes the absent hnswlib); the link leaves that one symbol unresolved (never called here).
//
// Protocol: ref_core <command> <input file> <output file>; binary little-endian files described per command.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "defs.h"       // reference header (via -I$(REFERENCE)/src)
#include "metis_io.h"   // reference header
#include "topn.h"       // reference header

namespace {
template <typename T>
std::vector<T> ReadVec(std::ifstream& in, size_t n) {
    std::vector<T> v(n);
    in.read(reinterpret_cast<char*>(v.data()), sizeof(T) * n);
    return v;
}
template <typename T>
T ReadOne(std::ifstream& in) {
    T x;
    in.read(reinterpret_cast<char*>(&x), sizeof(T));
    return x;
}
template <typename T>
void WriteOne(std::ofstream& out, T x) { out.write(reinterpret_cast<const char*>(&x), sizeof(T)); }
template <typename T>
void WriteVec(std::ofstream& out, const std::vector<T>& v) { out.write(reinterpret_cast<const char*>(v.data()), sizeof(T) * v.size()); }
}  // namespace

int main(int argc, char** argv) {
    if (argc != 4) {
        std::cerr << "usage: ref_core <command> <in> <out>" << std::endl;
        return 2;
    }
    std::string cmd = argv[1];
    if (cmd == "topn" || cmd == "topn_desc") {
        // in: u32 cases; per case: u64 n_items, u32 n, f32 dists[n_items], u32 ids[n_items]
        // out: per case: u32 count, f32 dists[count], u32 ids[count]   (Take(): ascending; topn_desc: ConvertTopKToNNVec)
        std::ifstream in(argv[2], std::ios::binary);
        std::ofstream out(argv[3], std::ios::binary);
        uint32_t cases = ReadOne<uint32_t>(in);
        for (uint32_t c = 0; c < cases; ++c) {
            uint64_t n_items = ReadOne<uint64_t>(in);
            uint32_t n = ReadOne<uint32_t>(in);
            auto dists = ReadVec<float>(in, n_items);
            auto ids = ReadVec<uint32_t>(in, n_items);
            TopN top(n);
            for (uint64_t i = 0; i < n_items; ++i) top.Add(std::make_pair(dists[i], ids[i]));
            NNVec res = cmd == "topn" ? top.Take() : ConvertTopKToNNVec(top);
            WriteOne<uint32_t>(out, (uint32_t)res.size());
            for (auto& x : res) WriteOne<float>(out, x.first);
            for (auto& x : res) WriteOne<uint32_t>(out, x.second);
        }
    } else if (cmd == "partition") {
        // in: u64 n, i32 partition[n]
        // out: i32 num_parts; clusters CSR (u32 S, u64 offsets[S+1], u32 ids[n]); cover (u64 n, then per point u32 cnt + i32 shards);
        //      remapped partition (i32[n]) after RemapPartitionIDs
        std::ifstream in(argv[2], std::ios::binary);
        std::ofstream out(argv[3], std::ios::binary);
        uint64_t n = ReadOne<uint64_t>(in);
        auto partition = ReadVec<int>(in, n);
        WriteOne<int32_t>(out, NumPartsInPartition(partition));
        Clusters clusters = ConvertPartitionToClusters(partition);
        WriteOne<uint32_t>(out, (uint32_t)clusters.size());
        uint64_t o = 0;
        WriteOne<uint64_t>(out, o);
        for (auto& c : clusters) { o += c.size(); WriteOne<uint64_t>(out, o); }
        for (auto& c : clusters) WriteVec(out, c);
        Cover cover = ConvertClustersToCover(clusters);
        WriteOne<uint64_t>(out, cover.size());
        for (auto& c : cover) { WriteOne<uint32_t>(out, (uint32_t)c.size()); WriteVec(out, c); }
        RemapPartitionIDs(partition);
        WriteVec(out, partition);
    } else if (cmd == "extract") {
        // in: u64 n, u32 d, f32 pts[n*d], u64 m, u32 bucket[m]   out: f32 [m*d] (ExtractPointsInBucket)
        std::ifstream in(argv[2], std::ios::binary);
        std::ofstream out(argv[3], std::ios::binary);
        PointSet points;
        points.n = ReadOne<uint64_t>(in);
        points.d = ReadOne<uint32_t>(in);
        points.coordinates = ReadVec<float>(in, points.n * points.d);
        uint64_t m = ReadOne<uint64_t>(in);
        auto bucket = ReadVec<uint32_t>(in, m);
        PointSet ps = ExtractPointsInBucket(bucket, points);
        WriteVec(out, ps.coordinates);
    } else if (cmd == "clusters_roundtrip") {
        // in: clusters text file; out: WriteClusters(ReadClusters(in))
        Clusters c = ReadClusters(argv[2]);
        WriteClusters(c, argv[3]);
    } else if (cmd == "clusters_read") {
        // in: clusters text file; out: u32 S, u64 offsets[S+1], u32 ids
        Clusters clusters = ReadClusters(argv[2]);
        std::ofstream out(argv[3], std::ios::binary);
        WriteOne<uint32_t>(out, (uint32_t)clusters.size());
        uint64_t o = 0;
        WriteOne<uint64_t>(out, o);
        for (auto& c : clusters) { o += c.size(); WriteOne<uint64_t>(out, o); }
        for (auto& c : clusters) WriteVec(out, c);
    } else if (cmd == "partition_roundtrip") {
        // in: METIS partition text; out: WriteMetisPartition(ReadMetisPartition(in))  (ReadMetisPartition compacts ids)
        auto p = ReadMetisPartition(argv[2]);
        WriteMetisPartition(p, argv[3]);
    } else {
        std::cerr << "unknown command " << cmd << std::endl;
        return 2;
    }
    return 0;
}
