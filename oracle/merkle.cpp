// ORACLE -- test infrastructure only (see gl.hpp header).  CPU restatement of the Ola fork's Merkle tree.
//
// Follows (relative to /root/reference/plonky2/plonky2/src/hash):
//   merkle_tree/mod.rs:180-201   new_v2: every leaf is hashed with hash_no_pad (never the <=4-element no-op)
//   merkle_tree/mod.rs:311-337   build_merkle_nodes: heap-ordered nodes, root at 1
//   merkle_tree/mod.rs:213-226   cap = nodes[2^h .. 2^(h+1))  (or the leaf hashes when the tree is all cap)
//   merkle_tree/mod.rs:228-259   digests re-laid per cap sub-tree (sibling pairs interleaved by layer)
//   merkle_tree/mod.rs:273-308   prove(): sibling lookup formula in that layout
//   merkle_proofs.rs:52-80       verify_merkle_proof_to_cap
// The hash is the configuration's Hasher (Poseidon, or Blake3_256 after set_hasher(HASH_BLAKE3), blake3.cpp).
#include "oracle.hpp"

namespace ola_oracle {

MerkleTree merkle_new_v2(std::vector<u64> leaves, size_t num_leaves, size_t leaf_len, int cap_height) {
    MerkleTree t;
    t.num_leaves = num_leaves;
    t.leaf_len = leaf_len;
    t.cap_height = cap_height;
    t.leaves = std::move(leaves);
    t.leaf_hash.resize(num_leaves);
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)num_leaves; i++) t.leaf_hash[i] = merkle_hash_leaf(t.get(i), leaf_len);
    size_t n = num_leaves / 2;
    t.nodes.assign(2 * n > 0 ? 2 * n : 1, HashOut{0, 0, 0, 0});
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; i++) t.nodes[n + i] = merkle_two_to_one(t.leaf_hash[2 * i], t.leaf_hash[2 * i + 1]);
    // level by level from the leaves' parents up to the root at 1 (a level's nodes are independent)
    for (size_t lvl = n / 2; lvl >= 1; lvl /= 2) {
#pragma omp parallel for schedule(static) if (lvl >= 64)
        for (long i = (long)lvl; i < (long)(2 * lvl); i++) t.nodes[i] = merkle_two_to_one(t.nodes[2 * i], t.nodes[2 * i + 1]);
    }
    size_t len_cap = (size_t)1 << cap_height;
    t.cap.resize(len_cap);
    for (size_t i = 0; i < len_cap; i++) t.cap[i] = (len_cap == num_leaves) ? t.leaf_hash[i] : t.nodes[len_cap + i];
    return t;
}

// Path of siblings from the leaf level up to (excluding) the cap level -- what the reference's prove() returns.
std::vector<HashOut> MerkleTree::prove(size_t leaf_index) const {
    int num_layers = log2_strict(num_leaves) - cap_height;
    std::vector<HashOut> sib;
    if (num_layers <= 0) return sib;
    sib.push_back(leaf_hash[leaf_index ^ 1]);
    size_t idx = (num_leaves + leaf_index) >> 1;  // heap index of the parent
    for (int l = 1; l < num_layers; l++) {
        sib.push_back(nodes[idx ^ 1]);
        idx >>= 1;
    }
    return sib;
}

std::vector<HashOut> merkle_reference_digests(const MerkleTree& t) {
    size_t leaves_len = t.num_leaves;
    int cap_height = t.cap_height;
    size_t num_digests = 2 * (leaves_len - ((size_t)1 << cap_height));
    std::vector<HashOut> digests(num_digests);
    int tree_height_sub_1 = log2_strict(leaves_len);
    int num_layers = tree_height_sub_1 - cap_height;
    size_t num_sub_tree_leaves = (size_t)1 << num_layers;
    size_t tree_len = num_digests >> cap_height;
    size_t num_trees = (size_t)1 << cap_height;
    if (num_digests == 0) return digests;
    for (size_t i = 0; i < num_trees; i++) {
        for (size_t pair = 0; pair < num_sub_tree_leaves; pair += 2) {
            size_t s = pair << 1;
            digests[tree_len * i + s] = t.leaf_hash[num_sub_tree_leaves * i + pair];
            digests[tree_len * i + s + 1] = t.leaf_hash[num_sub_tree_leaves * i + pair + 1];
        }
        for (int layer = 1; layer < num_layers; layer++) {
            size_t layer_nodes = num_sub_tree_leaves >> layer;
            for (size_t pair = 0; pair < layer_nodes; pair += 2) {
                size_t s = ((pair << layer) + ((size_t)1 << layer) - 1) << 1;
                size_t n_idx = ((size_t)1 << (tree_height_sub_1 - layer)) + layer_nodes * i + pair;
                digests[tree_len * i + s] = t.nodes[n_idx];
                digests[tree_len * i + s + 1] = t.nodes[n_idx + 1];
            }
        }
    }
    return digests;
}

std::vector<HashOut> merkle_prove_via_digests(const MerkleTree& t, const std::vector<HashOut>& digests,
                                              size_t leaf_index) {
    int cap_height = t.cap_height;
    int num_layers = log2_strict(t.num_leaves) - cap_height;
    size_t tree_index = leaf_index >> num_layers;
    size_t tree_len = digests.size() >> cap_height;
    const HashOut* dt = digests.data() + tree_len * tree_index;
    size_t pair_index = leaf_index & (((size_t)1 << num_layers) - 1);
    std::vector<HashOut> sib;
    for (int i = 0; i < num_layers; i++) {
        size_t parity = pair_index & 1;
        pair_index >>= 1;
        size_t siblings_index = (pair_index << (i + 1)) + ((size_t)1 << i) - 1;
        sib.push_back(dt[2 * siblings_index + (1 - parity)]);
    }
    return sib;
}

bool verify_merkle_proof_to_cap(const u64* leaf, size_t leaf_len, size_t leaf_index, const std::vector<HashOut>& cap,
                                const std::vector<HashOut>& siblings) {
    size_t index = leaf_index;
    HashOut cur = merkle_hash_leaf(leaf, leaf_len);
    for (const HashOut& s : siblings) {
        cur = (index & 1) ? merkle_two_to_one(s, cur) : merkle_two_to_one(cur, s);
        index >>= 1;
    }
    return index < cap.size() && cur == cap[index];
}

}  // namespace ola_oracle
