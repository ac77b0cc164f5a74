"""Optimizer step and LR schedule of the training loop.  Under the drop-in
overlay `segmentron.solver.optimizer` / `.lr_scheduler` resolve here; the rest of the reference's
solver package (losses) keeps coming from the reference checkout."""
