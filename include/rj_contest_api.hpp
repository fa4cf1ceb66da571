// rj_contest_api.hpp — the contest data/plan contract, as this repository's own
// declaration of the interface the reference publishes in include/plan.h and
// include/attribute.h (SIGMOD 2025 programming contest API).
//
// Why it exists: the GPU box receives only this repository, so tests of the C++
// drop-in shim (radix-join_amd/host/contest_execute.cpp) need the contract types
// without the reference tree.  It is name-compatible with the reference so that the
// very same shim source compiles inside the reference tree against the real
// <plan.h>.  Only what the boundary needs is declared:
//   DataType                         reference include/attribute.h:8-13
//   Page / Column / ColumnarTable    reference include/plan.h:54-105
//   ScanNode / JoinNode / PlanNode   reference include/plan.h:32-52
//   Plan (+ node/input builders)     reference include/plan.h:112-149
//   Contest::{build_context,destroy_context,execute}   reference include/plan.h:337-344
#pragma once
#include <cstddef>
#include <cstdint>
#include <tuple>
#include <utility>
#include <variant>
#include <vector>

enum class DataType { INT32, INT64, FP64, VARCHAR };

constexpr size_t PAGE_SIZE = 8192;
struct alignas(8) Page {
    std::byte data[PAGE_SIZE];
};

// Owns its pages: every pages[i] is a `new Page`, deleted by the destructor.
struct Column {
    DataType           type;
    std::vector<Page*> pages;

    explicit Column(DataType t) : type(t) {}
    Column(const Column&) = delete;
    Column& operator=(const Column&) = delete;
    Column(Column&& o) noexcept : type(o.type), pages(std::move(o.pages)) { o.pages.clear(); }
    Column& operator=(Column&& o) noexcept {
        if (this != &o) {
            release();
            type = o.type;
            pages = std::move(o.pages);
            o.pages.clear();
        }
        return *this;
    }
    ~Column() { release(); }
    Page* new_page() {
        pages.push_back(new Page);
        return pages.back();
    }

   private:
    void release() {
        for (Page* p : pages) delete p;
        pages.clear();
    }
};

struct ColumnarTable {
    size_t              num_rows{0};
    std::vector<Column> columns;
};

struct ScanNode {
    size_t base_table_id;
};
struct JoinNode {
    bool   build_left;
    size_t left, right;
    size_t left_attr, right_attr;
};
struct PlanNode {
    std::variant<ScanNode, JoinNode>          data;
    std::vector<std::tuple<size_t, DataType>> output_attrs;
    PlanNode(std::variant<ScanNode, JoinNode> d, std::vector<std::tuple<size_t, DataType>> o)
        : data(std::move(d)), output_attrs(std::move(o)) {}
};

struct Plan {
    std::vector<PlanNode>      nodes;
    std::vector<ColumnarTable> inputs;
    size_t                     root{0};

    size_t new_join_node(bool build_left, size_t left, size_t right, size_t left_attr,
                         size_t right_attr, std::vector<std::tuple<size_t, DataType>> out) {
        nodes.emplace_back(JoinNode{build_left, left, right, left_attr, right_attr}, std::move(out));
        return nodes.size() - 1;
    }
    size_t new_scan_node(size_t base_table_id, std::vector<std::tuple<size_t, DataType>> out) {
        nodes.emplace_back(ScanNode{base_table_id}, std::move(out));
        return nodes.size() - 1;
    }
    size_t new_input(ColumnarTable t) {
        inputs.emplace_back(std::move(t));
        return inputs.size() - 1;
    }
};

namespace Contest {
void*         build_context();
void          destroy_context(void*);
ColumnarTable execute(const Plan& plan, void* context);
}  // namespace Contest
