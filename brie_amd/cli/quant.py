"""brie-quant on MI355X: same flags, defaults and outputs as the reference CLI
(/root/reference/brie/bin/quant.py:13-219), driving brie_amd.fitBRIE.

    python -m brie_amd.cli.quant -i counts.npz -c cells.tsv -o out/brie_quant.h5ad --LRTindex=0
    torchrun --nproc-per-node 8 -m brie_amd.cli.quant ...      # genes sharded over 8 GPUs (RCCL)

Differences: `--nproc` is accepted and ignored (it only sized TensorFlow's CPU thread pools,
quant.py:207-211); new `--seed` and `--device`; without the `anndata` package the fitted object is
written as an .npz bundle instead of .h5ad (brie_amd/io.py) -- the `.brie_ident.tsv` table is identical.
"""
import os
import sys
from optparse import OptionGroup, OptionParser

import numpy as np


def _read_table(path):
    delim = "," if path.endswith('csv') or path.endswith('csv.gz') else "\t"      # quant.py:48-52
    return np.genfromtxt(path, dtype="str", delimiter=delim)


def quant(in_file, cell_file=None, gene_file=None, out_file=None, LRT_index=[],
          layer_keys=['isoform1', 'isoform2', 'ambiguous'], intercept=None, intercept_mode='gene', nproc=1,
          min_counts=50, min_counts_uniq=10, min_cells_uniq=30, min_MIF_uniq=0.001, min_iter=5000,
          max_iter=20000, MC_size=1, batch_size=500000, pseudo_count=0.01, base_mode='full',
          seed=0, device=0, comm=None, fit_function=None, **fit_kwargs):
    """Quantify splicing isoforms and detect variable splicing events (quant.py:13-130)."""
    from .. import io as bio
    from ..preprocessing import filter_genes, match
    from ..version import __version__
    if out_file is None:
        print("No given out_file, use the dir for input file.")
        out_file = os.path.dirname(os.path.abspath(in_file)) + "/brie_quant.h5ad"
    os.makedirs(os.path.dirname(os.path.abspath(out_file)), exist_ok=True)

    if in_file.endswith(".h5ad"):
        adata = bio.read_h5ad(in_file)
    elif in_file.endswith(".npz"):
        adata = bio.read_npz(in_file)
    else:
        raise ValueError("input must be .h5ad or .npz: %s" % in_file)

    Xc = Xc_ids = None
    if cell_file is not None:                                                     # quant.py:47-63
        tab = _read_table(cell_file)
        idx = match(list(adata.obs.index), list(tab[1:, 0]))
        hit = np.array([i is not None for i in idx])
        rows = idx[hit].astype(int)
        print("[BRIE2] %.1f%% cells are matched with features" % (np.mean(hit) * 100))
        Xc = tab[rows + 1, 1:].astype(np.float32)
        Xc_ids = tab[0, 1:]
        adata = adata[hit, :]

    print("layers:", layer_keys)
    adata = filter_genes(adata, min_counts=min_counts, min_counts_uniq=min_counts_uniq,
                         min_cells_uniq=min_cells_uniq, min_MIF_uniq=min_MIF_uniq,
                         uniq_layers=layer_keys[:2], ambg_layers=layer_keys[2:], copy=True)   # quant.py:70-75

    Xg = Xg_ids = None
    if gene_file is not None:                                                     # quant.py:78-95
        tab = _read_table(gene_file)
        idx = match(list(adata.var.index), list(tab[1:, 0]))
        hit = np.array([i is not None for i in idx])
        rows = idx[hit].astype(int)
        print("[BRIE2] %.1f%% genes are matched with features" % (np.mean(hit) * 100))
        Xg = tab[rows + 1, 1:].astype(np.float32)
        Xg_ids = tab[0, 1:]
        adata = adata[:, hit]
    print(adata)
    tau_prior = [1, 1] if 'unspliced' in adata.layers else [3, 27]                 # quant.py:102-105 (unused downstream)

    if fit_function is None:
        from ..models import fitBRIE as fit_function
    fit_function(adata, Xc=Xc, Xg=Xg, LRT_index=LRT_index, layer_keys=layer_keys, intercept=intercept,
                 intercept_mode=intercept_mode, min_iter=min_iter, max_iter=max_iter, MC_size=MC_size,
                 batch_size=batch_size, pseudo_count=pseudo_count, base_mode=base_mode, tau_prior=tau_prior,
                 seed=seed, device=device, comm=comm, **fit_kwargs)
    adata.uns['brie_version'] = __version__
    adata.uns['Xc_ids'] = Xc_ids
    adata.uns['Xg_ids'] = Xg_ids

    if comm is None or comm.rank == 0:
        written = bio.write_results(adata, out_file)
        table = ".".join(out_file.split('.')[:-1]) + '.brie_ident.tsv'            # quant.py:127-130
        bio.dump_results(adata).to_csv(table, sep='\t', header=True, index=True, index_label='GeneID',
                                       float_format='%.3e')
        print("[BRIE2] results: %s, %s" % (written, table))
    return adata


def build_parser():
    parser = OptionParser()
    parser.add_option("--inFile", "-i", dest="in_file", default=None,
                      help="Input read count matrices in AnnData h5ad or brie npz format.")
    parser.add_option("--cellFile", "-c", dest="cell_file", default=None,
                      help="File for cell features in tsv[.gz] with cell and feature ids.")
    parser.add_option("--geneFile", "-g", dest="gene_file", default=None,
                      help="File for gene features in tsv[.gz] with gene and feature ids.")
    parser.add_option("--out_file", "-o", dest="out_file", default=None,
                      help="Full path of output file for annData in h5ad [default: $inFile/brie_quant.h5ad]")
    parser.add_option("--LRTindex", dest="LRT_index", default="None",
                      help="Index (0-based) of cell features to test with LRT: All, None or comma separated "
                           "integers [default: %default]")
    parser.add_option("--testBase", dest="test_base", default="full",
                      help="Features in testing base model: full, null  [default: %default]")
    parser.add_option("--interceptMode", dest="intercept_mode", default="None",
                      help="Intercept mode: gene, cell or None [default: %default]")
    parser.add_option("--layers", dest="layers", default="isoform1,isoform2,ambiguous",
                      help="Comma separated layers two or three for estimating Psi [default: %default]")
    parser.add_option("--seed", type="int", dest="seed", default=0, help="Seed of the noise stream [default: %default]")
    parser.add_option("--device", type="int", dest="device", default=None,
                      help="GPU ordinal [default: LOCAL_RANK or 0]")
    g1 = OptionGroup(parser, "Gene filtering")
    g1.add_option("--minCount", type="int", dest="min_count", default=50,
                  help="Minimum total counts for fitltering genes [default: %default]")
    g1.add_option("--minUniqCount", type="int", dest="min_uniq_count", default=10,
                  help="Minimum unique counts for fitltering genes [default: %default]")
    g1.add_option("--minCell", type="int", dest="min_cell", default=30,
                  help="Minimum number of cells with unique count for fitltering genes [default: %default]")
    g1.add_option("--minMIF", type="float", dest="min_MIF", default=0.001,
                  help="Minimum minor isoform frequency in unique count [default: %default]")
    g2 = OptionGroup(parser, "VI Optimization")
    g2.add_option("--MCsize", type="int", dest="MC_size", default=3,
                  help="Sample size for Monte Carlo Expectation [default: %default]")
    g2.add_option("--minIter", type="int", dest="min_iter", default=5000,
                  help="Minimum number of iterations [default: %default]")
    g2.add_option("--maxIter", type="int", dest="max_iter", default=20000,
                  help="Maximum number of iterations [default: %default]")
    g2.add_option("--batchSize", type="int", dest="batch_size", default=500000,
                  help="Element size per batch: n_gene * total cell [default: %default] (only with emulated batches)")
    g2.add_option("--pseudoCount", type="float", dest="pseudo_count", default=0.01,
                  help="Pseudo count to add on unique count matrices [default: %default]")
    g2.add_option("--nproc", "-p", type="int", dest="nproc", default=6,
                  help="Ignored (TensorFlow CPU threads in the reference) [default: %default]")
    parser.add_option_group(g1)
    parser.add_option_group(g2)
    return parser


def parse_lrt_index(text):
    """quant.py:198-203."""
    if text.upper() == "NONE":
        return []
    if text.upper() == "ALL":
        return None
    return np.array(text.split(","), float).astype(int)


def main(argv=None):
    from ..version import __version__
    argv = sys.argv[1:] if argv is None else argv
    parser = build_parser()
    options, _ = parser.parse_args(argv)
    if len(argv) == 0:
        print("Welcome to brie-quant (brie_amd v%s, MI355X)!\n" % __version__)
        print("use -h or --help for help on argument.")
        sys.exit(1)
    if options.in_file is None:
        print("[BRIE2] Error: need --inFile for count matrices (h5ad or npz).")
        sys.exit(1)
    LRT_index = parse_lrt_index(options.LRT_index)
    intercept = None if options.intercept_mode.upper() in ["GENE", "CELL"] else 0   # quant.py:205

    comm = None
    device = options.device
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:        # one process per GPU
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        device = local if device is None else device
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        from ..sharding import GeneComm
        comm = GeneComm(device=torch.device("cuda", device))
    device = 0 if device is None else device

    quant(options.in_file, options.cell_file, options.gene_file, options.out_file, LRT_index,
          options.layers.split(','), intercept, options.intercept_mode, options.nproc, options.min_count,
          options.min_uniq_count, options.min_cell, options.min_MIF, options.min_iter, options.max_iter,
          options.MC_size, options.batch_size, options.pseudo_count, options.test_base,
          seed=options.seed, device=device, comm=comm)
    if comm is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
