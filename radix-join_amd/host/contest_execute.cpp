// contest_execute.cpp — drop-in replacement for the reference's src/execute.cpp.
//
// Defines the three functions the harness links against (reference include/plan.h:337-344,
// reference definitions src/execute.cpp:316-330) on top of the C-ABI in include/rj.h:
//   Contest::build_context()    -> rj_context_create   (HIP stream, HBM cache, pinned staging)
//   Contest::execute(plan, ctx) -> flatten Plan to rj_plan (PODs + page pointers),
//                                  rj_execute, copy the result into `new Page`s so that
//                                  Column::~Column (plan.h:95-99) can delete them
//   Contest::destroy_context()  -> rj_context_destroy
// Only public members of the contract types are touched, so this file compiles unchanged
// against the reference's real <plan.h> and against this repository's
// include/contest_compat/plan.h.  Errors surface as std::runtime_error, like the reference's
// (src/execute.cpp:280, build_table.cpp:335).
#include <plan.h>

#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

#include "rj.h"

namespace Contest {

namespace {
[[noreturn]] void fail(rj_context* ctx) {
    const char* m = rj_last_error(ctx);
    throw std::runtime_error(m && *m ? m : "radix-join: unknown error");
}
}  // namespace

// RJ_DEVICES = "all" or a comma-separated list of HIP ordinals ("0,1,2,3"): the context owns
// those GPUs and rj_execute shards every JoinNode it can across them (one all-to-all over xGMI
// per relation); unset = the current device only.  Nothing else changes for the harness.
void* build_context() {
    rj_context*          ctx = nullptr;
    rj_config            cfg{};
    std::vector<int32_t> devs;
    cfg.device = -1;
    cfg.flags = RJ_CTX_PREWARM;  // set-up cost lands here, not in the first query's execute()
    if (const char* e = std::getenv("RJ_DEVICES")) {
        std::string v(e);
        if (v == "all") {
            rj_context* probe = nullptr;
            rj_config   one{};
            one.device = -1;
            if (rj_context_create(&probe, &one) != RJ_OK) fail(nullptr);
            rj_device_info info{};
            (void)rj_device_query(probe, &info);
            rj_context_destroy(probe);
            for (int32_t d = 0; d < info.device_count; ++d) devs.push_back(d);
        } else {
            size_t pos = 0;
            while (pos < v.size()) {
                size_t end = v.find(',', pos);
                if (end == std::string::npos) end = v.size();
                if (end > pos) devs.push_back(std::atoi(v.substr(pos, end - pos).c_str()));
                pos = end + 1;
            }
        }
        // a sharded join wants a power-of-two number of ranks
        size_t n = 1;
        while (n * 2 <= devs.size()) n *= 2;
        devs.resize(devs.empty() ? 0 : n);
    }
    if (devs.size() > 1) {
        cfg.n_devices = (int32_t)devs.size();
        cfg.devices = devs.data();
    } else if (devs.size() == 1) {
        cfg.device = devs[0];
    }
    if (rj_context_create(&ctx, &cfg) != RJ_OK) fail(nullptr);
    return ctx;
}

void destroy_context(void* context) { rj_context_destroy(static_cast<rj_context*>(context)); }

ColumnarTable execute(const Plan& plan, void* context) {
    auto* ctx = static_cast<rj_context*>(context);
    if (!ctx) throw std::runtime_error("radix-join: execute() needs the context of build_context()");

    // ---- flatten the Plan (no copies of page data: pointers only)
    std::vector<rj_node>               nodes(plan.nodes.size());
    std::vector<std::vector<uint64_t>> out_idx(plan.nodes.size());
    std::vector<std::vector<int32_t>>  out_type(plan.nodes.size());
    for (size_t i = 0; i < plan.nodes.size(); ++i) {
        const PlanNode& n = plan.nodes[i];
        rj_node&        d = nodes[i];
        d = rj_node{};
        for (const auto& [idx, type] : n.output_attrs) {
            out_idx[i].push_back(idx);
            out_type[i].push_back(static_cast<int32_t>(type));
        }
        d.n_out = out_idx[i].size();
        d.out_idx = out_idx[i].data();
        d.out_type = out_type[i].data();
        if (const auto* j = std::get_if<JoinNode>(&n.data)) {
            d.kind = RJ_NODE_JOIN;
            d.build_left = j->build_left ? 1 : 0;
            d.left = j->left;
            d.right = j->right;
            d.left_attr = j->left_attr;
            d.right_attr = j->right_attr;
        } else {
            d.kind = RJ_NODE_SCAN;
            d.base_table_id = std::get<ScanNode>(n.data).base_table_id;
        }
    }
    std::vector<rj_input>               inputs(plan.inputs.size());
    std::vector<std::vector<rj_column>> cols(plan.inputs.size());
    for (size_t t = 0; t < plan.inputs.size(); ++t) {
        const ColumnarTable& in = plan.inputs[t];
        for (const Column& c : in.columns) {
            rj_column rc{};
            rc.type = static_cast<int32_t>(c.type);
            rc.n_pages = c.pages.size();
            // Page is a standard-layout 8192-byte block: Page* is the page's address
            rc.pages = reinterpret_cast<const void* const*>(c.pages.data());
            cols[t].push_back(rc);
        }
        inputs[t].num_rows = in.num_rows;
        inputs[t].n_cols = cols[t].size();
        inputs[t].cols = cols[t].data();
    }
    rj_plan p{};
    p.n_nodes = nodes.size();
    p.nodes = nodes.data();
    p.n_inputs = inputs.size();
    p.inputs = inputs.data();
    p.root = plan.root;

    rj_result* res = nullptr;
    if (rj_execute(ctx, &p, &res) != RJ_OK) fail(ctx);

    // ---- hand the result over as `new Page`s
    ColumnarTable out;
    try {
        out.num_rows = rj_result_num_rows(res);
        const uint64_t nc = rj_result_num_cols(res);
        for (uint64_t c = 0; c < nc; ++c) {
            out.columns.emplace_back(static_cast<DataType>(rj_result_col_type(res, c)));
            Column&        col = out.columns.back();
            const uint64_t np = rj_result_col_pages(res, c);
            col.pages.reserve(np);
            for (uint64_t i = 0; i < np; ++i) col.new_page();
            if (np && rj_result_copy_pages(res, c, reinterpret_cast<void* const*>(col.pages.data()),
                                           np) != RJ_OK)
                fail(ctx);
        }
    } catch (...) {
        rj_result_free(res);
        throw;
    }
    rj_result_free(res);
    return out;
}

}  // namespace Contest
