// heap.h — BestHeap: the host-side mirror of the reference's BestAssociationsHeap
// (src/best_associations_heap.h:32-54, src/best_associations_heap.cpp:26-127).
//
// The reference keeps (k-mer, score, row) tuples in a std::priority_queue ordered by
// cmp(a, b) = a.score > b.score (src/kmer_general.h:113-128). Which of several equal-score
// entries survives at the boundary, and the order equal scores pop in (= the rank written
// into the .bim names), is decided by libstdc++'s push_heap / pop_heap and the exact history
// of effective pushes. We therefore use the same standard container over the same tuple and
// comparator shapes and replay exactly the effective pushes, in row order.
#pragma once
#include <stdint.h>

#include <algorithm>
#include <queue>
#include <tuple>
#include <vector>

namespace kgwas {

typedef std::tuple<uint64_t, double, size_t> HeapItem;  // k-mer, score, row
struct HeapItemGreater {
    inline bool operator()(const HeapItem& l, const HeapItem& r) const { return std::get<1>(l) > std::get<1>(r); }
};
typedef std::priority_queue<HeapItem, std::vector<HeapItem>, HeapItemGreater> HeapQueue;

class BestHeap {
   public:
    explicit BestHeap(size_t max_results) : n_res_(max_results), inserted_(0), pushes_(0), lowest_(0) {}

    // add_association (src/best_associations_heap.cpp:43-59). Returns true if the heap changed.
    inline bool add(uint64_t kmer, double score, size_t row) {
        inserted_++;
        if (q_.size() < n_res_) {
            q_.push(HeapItem(kmer, score, row));
            pushes_++;
            lowest_ = std::get<1>(q_.top());
            return true;
        }
        if (score > lowest_) {
            HeapItem e(kmer, score, row);
            q_.pop();
            q_.push(e);
            pushes_++;
            lowest_ = std::get<1>(q_.top());
            return true;
        }
        return false;
    }
    inline bool full() const { return q_.size() >= n_res_; }
    inline size_t size() const { return q_.size(); }
    inline size_t capacity() const { return n_res_; }
    inline double lowest() const { return lowest_; }
    inline uint64_t inserted() const { return inserted_; }
    inline uint64_t pushes() const { return pushes_; }
    // rows skipped on the device still count as insertions (cnt_kmers++ happens for every row)
    inline void count_skipped(uint64_t n) { inserted_ += n; }

    // output_to_file_with_scores order (:82-92): ascending pops from a copy.
    void pop_all(std::vector<uint64_t>& kmer, std::vector<double>& score, std::vector<uint64_t>& row) const {
        HeapQueue tmp(q_);
        kmer.clear();
        score.clear();
        row.clear();
        while (!tmp.empty()) {
            kmer.push_back(std::get<0>(tmp.top()));
            score.push_back(std::get<1>(tmp.top()));
            row.push_back(std::get<2>(tmp.top()));
            tmp.pop();
        }
    }

   private:
    size_t n_res_;
    HeapQueue q_;
    uint64_t inserted_, pushes_;
    double lowest_;
};

}  // namespace kgwas
