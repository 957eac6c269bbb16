// heap.h — BestHeap: the host-side mirror of the reference's BestAssociationsHeap
// (src/best_associations_heap.h:32-54, src/best_associations_heap.cpp:26-127).
//
// The reference keeps (k-mer, score, row) tuples in a std::priority_queue ordered by
// cmp(a, b) = a.score > b.score (src/kmer_general.h:113-128). Which of several equal-score
// entries survives at the boundary, and the order equal scores pop in (= the rank written
// into the .bim names), is decided by libstdc++'s push_heap / pop_heap and the exact history
// of effective pushes. std::priority_queue is nothing but a vector driven by exactly these calls
//     push: c.push_back(x); std::push_heap(c.begin(), c.end(), comp);
//     pop : std::pop_heap(c.begin(), c.end(), comp); c.pop_back();
//     top : c.front()
// and the algorithms only ever look at the elements through `comp`. We issue the same calls on a
// vector of 16-byte (score, slot) entries — k-mer and row live in side arrays indexed by slot — so
// every comparison, and therefore every move, is the one the reference's 24-byte tuples would
// make, at 2/3 of the memory traffic of the replay (the hot loop of the host side).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace kgwas {

class BestHeap {
    struct Ent {
        double score;
        uint32_t slot;
    };
    struct Greater {
        inline bool operator()(const Ent& l, const Ent& r) const { return l.score > r.score; }
    };

   public:
    explicit BestHeap(size_t max_results) : n_res_(max_results), inserted_(0), pushes_(0), lowest_(0) {}

    // add_association (src/best_associations_heap.cpp:43-59). Returns true if the heap changed.
    inline bool add(uint64_t kmer, double score, size_t row) {
        inserted_++;
        if (v_.size() < n_res_) {
            const uint32_t slot = (uint32_t)v_.size();
            kmer_.push_back(kmer);
            row_.push_back(row);
            v_.push_back(Ent{score, slot});
            std::push_heap(v_.begin(), v_.end(), Greater());
            pushes_++;
            lowest_ = v_.front().score;
            return true;
        }
        if (score > lowest_) {
            std::pop_heap(v_.begin(), v_.end(), Greater());
            const uint32_t slot = v_.back().slot;  // the evicted minimum's slot is reused
            v_.pop_back();
            kmer_[slot] = kmer;
            row_[slot] = row;
            v_.push_back(Ent{score, slot});
            std::push_heap(v_.begin(), v_.end(), Greater());
            pushes_++;
            lowest_ = v_.front().score;
            return true;
        }
        return false;
    }
    inline bool full() const { return v_.size() >= n_res_; }
    inline size_t size() const { return v_.size(); }
    inline size_t capacity() const { return n_res_; }
    inline double lowest() const { return lowest_; }
    inline uint64_t inserted() const { return inserted_; }
    inline uint64_t pushes() const { return pushes_; }

    // output_to_file_with_scores order (:82-92): ascending pops from a copy.
    void pop_all(std::vector<uint64_t>& kmer, std::vector<double>& score, std::vector<uint64_t>& row) const {
        std::vector<Ent> tmp(v_);
        kmer.clear();
        score.clear();
        row.clear();
        while (!tmp.empty()) {
            kmer.push_back(kmer_[tmp.front().slot]);
            score.push_back(tmp.front().score);
            row.push_back(row_[tmp.front().slot]);
            std::pop_heap(tmp.begin(), tmp.end(), Greater());
            tmp.pop_back();
        }
    }

   private:
    size_t n_res_;
    std::vector<Ent> v_;
    std::vector<uint64_t> kmer_, row_;
    uint64_t inserted_, pushes_;
    double lowest_;
};

}  // namespace kgwas
