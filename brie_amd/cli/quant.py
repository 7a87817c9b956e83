"""brie-quant on MI355X.

Command-line front end with the flag names and default values of the reference tool
(/root/reference/brie/bin/quant.py:138-187) driving `brie_amd.fitBRIE`:

    python -m brie_amd.cli.quant -i counts.npz -c cells.tsv -o out/brie_quant.h5ad --LRTindex=0
    torchrun --nproc-per-node 8 -m brie_amd.cli.quant ...      # genes sharded over 8 GPUs (RCCL)

What differs from the reference tool: `--nproc` is parsed and ignored (it sized TensorFlow's CPU
thread pools, quant.py:207-211); `--seed` / `--device` are new; when the `anndata` package is not
installed the fitted object is stored as an .npz bundle instead of .h5ad (brie_amd/io.py).  The result
table `<out>.brie_ident.tsv` has the reference's columns.
"""
import argparse
import os
import sys

import numpy as np

#: (flag, short, dest, type, default, help) -- names and defaults follow quant.py:138-187
_OPTIONS = [
    ("--inFile", "-i", "in_file", str, None, "count matrices: AnnData .h5ad or brie .npz"),
    ("--cellFile", "-c", "cell_file", str, None, "table of cell features (tsv/csv[.gz]: first column cell id, header row)"),
    ("--geneFile", "-g", "gene_file", str, None, "table of gene features (tsv/csv[.gz]: first column gene id, header row)"),
    ("--out_file", "-o", "out_file", str, None, "output path [default: brie_quant.h5ad next to the input]"),
    ("--LRTindex", None, "LRT_index", str, "None", "0-based cell-feature columns to test: All, None or e.g. 0,2"),
    ("--testBase", None, "test_base", str, "full", "base model of the test: full or null"),
    ("--interceptMode", None, "intercept_mode", str, "None", "gene, cell or None (= intercept fixed at 0)"),
    ("--layers", None, "layers", str, "isoform1,isoform2,ambiguous", "two or three count layers, comma separated"),
    ("--minCount", None, "min_count", int, 50, "gene filter: minimum total counts"),
    ("--minUniqCount", None, "min_uniq_count", int, 10, "gene filter: minimum isoform-specific counts"),
    ("--minCell", None, "min_cell", int, 30, "gene filter: minimum cells with isoform-specific counts"),
    ("--minMIF", None, "min_MIF", float, 0.001, "gene filter: minimum minor-isoform frequency"),
    ("--MCsize", None, "MC_size", int, 3, "Monte-Carlo samples per ELBO evaluation"),
    ("--minIter", None, "min_iter", int, 5000, "iterations of the staged schedule"),
    ("--maxIter", None, "max_iter", int, 20000, "iteration cap of the convergence loop"),
    ("--batchSize", None, "batch_size", int, 500000, "elements (genes x cells) per convergence batch"),
    ("--pseudoCount", None, "pseudo_count", float, 0.01, "added to both isoform-specific layers where they have reads"),
    ("--nproc", "-p", "nproc", int, 6, "ignored (CPU threads of the TensorFlow reference)"),
    ("--seed", None, "seed", int, 0, "seed of the noise stream"),
    ("--commonNoise", None, "common_noise", int, 0, "1: base and test fits of the LRT share the noise stream "
                                                     "(common random numbers: far less Monte-Carlo error in ELBO_gain)"),
    ("--device", None, "device", int, None, "GPU ordinal [default: LOCAL_RANK or 0]"),
]


def build_parser():
    parser = argparse.ArgumentParser(prog="brie-quant", description=__doc__.split("\n\n")[0])
    for flag, short, dest, typ, default, text in _OPTIONS:
        names = [flag] + ([short] if short else [])
        parser.add_argument(*names, dest=dest, type=typ, default=default, help="%s [default: %s]" % (text, default))
    return parser


def parse_lrt_index(text):
    """'None' -> no test, 'All' -> every feature, '0,2' -> those columns (quant.py:198-203)."""
    key = text.strip().upper()
    if key == "NONE":
        return []
    if key == "ALL":
        return None
    return np.array(text.split(","), float).astype(int)


def _load_table(path):
    sep = "," if path.endswith(("csv", "csv.gz")) else "\t"
    return np.genfromtxt(path, dtype="str", delimiter=sep)


def _match_features(ids, table, what):
    """Rows of `table` (first column = id) in the order of `ids`; returns (hit mask, float32 matrix, names)."""
    from ..preprocessing import match
    where = match(list(ids), list(table[1:, 0]))
    hit = np.array([w is not None for w in where], dtype=bool)
    rows = np.array([w for w in where if w is not None], dtype=int)
    print("[BRIE2] %.1f%% %s are matched with features" % (100.0 * hit.mean(), what))
    return hit, table[rows + 1, 1:].astype(np.float32), table[0, 1:]


def quant(in_file, cell_file=None, gene_file=None, out_file=None, LRT_index=[],
          layer_keys=['isoform1', 'isoform2', 'ambiguous'], intercept=None, intercept_mode='gene', nproc=1,
          min_counts=50, min_counts_uniq=10, min_cells_uniq=30, min_MIF_uniq=0.001, min_iter=5000,
          max_iter=20000, MC_size=1, batch_size=500000, pseudo_count=0.01, base_mode='full',
          seed=0, device=0, comm=None, fit_function=None, **fit_kwargs):
    """Load counts, align feature tables, filter genes, fit, write results (the steps of quant.py:13-130)."""
    from .. import io as bio
    from ..preprocessing import filter_genes
    from ..version import __version__
    in_abs = os.path.abspath(in_file)
    if out_file is None:
        out_file = os.path.join(os.path.dirname(in_abs), "brie_quant.h5ad")
        print("No given out_file, use the dir for input file.")
    os.makedirs(os.path.dirname(os.path.abspath(out_file)), exist_ok=True)

    reader = {".h5ad": bio.read_h5ad, ".npz": bio.read_npz}.get(os.path.splitext(in_file)[1])
    if reader is None:
        raise ValueError("input must be .h5ad or .npz: %s" % in_file)
    adata = reader(in_file)

    Xc = Xc_ids = Xg = Xg_ids = None
    if cell_file is not None:                       # cells without features are dropped (quant.py:47-63)
        hit, Xc, Xc_ids = _match_features(adata.obs.index, _load_table(cell_file), "cells")
        adata = adata[hit, :]
    print("layers:", layer_keys)
    filter_genes(adata, min_counts=min_counts, min_counts_uniq=min_counts_uniq,          # in place, as quant.py:70-75
                 min_cells_uniq=min_cells_uniq, min_MIF_uniq=min_MIF_uniq,
                 uniq_layers=layer_keys[:2], ambg_layers=layer_keys[2:])
    if gene_file is not None:                       # likewise for genes (quant.py:78-95)
        hit, Xg, Xg_ids = _match_features(adata.var.index, _load_table(gene_file), "genes")
        adata = adata[:, hit]
    print(adata)
    tau_prior = [1, 1] if 'unspliced' in adata.layers else [3, 27]      # accepted, unused downstream (quant.py:102-105)

    if fit_function is None:
        from ..models import fitBRIE as fit_function
    fit_function(adata, Xc=Xc, Xg=Xg, LRT_index=LRT_index, layer_keys=layer_keys, intercept=intercept,
                 intercept_mode=intercept_mode, min_iter=min_iter, max_iter=max_iter, MC_size=MC_size,
                 batch_size=batch_size, pseudo_count=pseudo_count, base_mode=base_mode, tau_prior=tau_prior,
                 seed=seed, device=device, comm=comm, **fit_kwargs)
    adata.uns.update(brie_version=__version__, Xc_ids=Xc_ids, Xg_ids=Xg_ids)

    if comm is None or comm.rank == 0:
        written = bio.write_results(adata, out_file)
        table = os.path.splitext(out_file)[0] + '.brie_ident.tsv'
        bio.dump_results(adata).to_csv(table, sep='\t', header=True, index=True, index_label='GeneID',
                                       float_format='%.3e')
        print("[BRIE2] results: %s, %s" % (written, table))
    return adata


def _distributed_comm(device):
    """One process per GPU under torchrun: NCCL (= RCCL) group + GeneComm; None for a single process."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) <= 1:
        return None, (0 if device is None else device)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # RCCL across processes: dmabuf IPC
    import torch
    import torch.distributed as dist
    from ..sharding import GeneComm
    device = int(os.environ.get("LOCAL_RANK", "0")) if device is None else device
    torch.cuda.set_device(device)
    dist.init_process_group("nccl", device_id=torch.device("cuda", device))
    return GeneComm(device=torch.device("cuda", device)), device


def main(argv=None):
    from ..version import __version__
    argv = sys.argv[1:] if argv is None else list(argv)
    if not argv:
        print("Welcome to brie-quant (brie_amd v%s, MI355X)!\n" % __version__)
        print("use -h or --help for help on argument.")
        sys.exit(1)
    opt = build_parser().parse_args(argv)
    if opt.in_file is None:
        print("[BRIE2] Error: need --inFile for count matrices (h5ad or npz).")
        sys.exit(1)
    # only 'gene' / 'cell' learn an intercept; anything else pins it to 0 (quant.py:205)
    intercept = None if opt.intercept_mode.upper() in ("GENE", "CELL") else 0
    comm, device = _distributed_comm(opt.device)
    quant(opt.in_file, opt.cell_file, opt.gene_file, opt.out_file, parse_lrt_index(opt.LRT_index),
          opt.layers.split(','), intercept, opt.intercept_mode, opt.nproc, opt.min_count, opt.min_uniq_count,
          opt.min_cell, opt.min_MIF, opt.min_iter, opt.max_iter, opt.MC_size, opt.batch_size,
          opt.pseudo_count, opt.test_base, seed=opt.seed, device=device, comm=comm,
          common_noise=bool(opt.common_noise))
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
