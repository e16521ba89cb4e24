"""Pipelined frame stream on the C-ABI -- the Python twin of lzb_vio::System::StreamPush / StreamPoll
(host/src/System.cpp), which put a queue behind the reference's Step_ros entry (src/System.cpp:60-74).

Frames arrive one at a time (host memory).  They are gathered into micro-batches of `depth` pairs in page-locked
memory; a full micro-batch is uploaded and launched without waiting (svo_upload_frames + svo_track_uploaded_async,
at most two in flight, the pose chain continues on the device), and its records are picked up when they are ready
(svo_results_ready / svo_collect_results).  Poses come back `depth` to 3 x `depth` frames late and are byte for byte
those of the per-frame loop (svo_add_frame): consecutive pairs are independent."""
import numpy as np


class FrameStream:
    def __init__(self, ctx, depth):
        assert depth >= 1 and depth <= ctx.cfg.max_batch
        self.ctx, self.depth = ctx, int(depth)
        self.w, self.h = ctx.width, ctx.height
        self.pitch = (self.w + 255) // 256 * 256
        self.pin = [[ctx.host_frames(self.depth + 1, self.pitch) for _cam in range(2)] for _buf in range(2)]
        self.buf, self.n, self.chunk = 0, 0, 0
        self.outstanding = []              # pairs of the micro-batches in flight, oldest first
        self.uploaded = [False, False]

    def restart(self):
        """A new sequence on the same context (nothing may be in flight): frame 0 next, pose chain from identity."""
        assert not self.outstanding
        self.buf, self.n, self.chunk = 0, 0, 0

    def close(self):
        for b in self.pin:
            for v in b:
                self.ctx.host_free(v)
        self.pin = None

    def _collect(self):
        return self.ctx.collect_results(self.outstanding.pop(0))

    def _submit(self):
        done = []
        if self.n < 2:
            return done
        if len(self.outstanding) == 2:
            done.append(self._collect())
        b = self.buf
        # after the first micro-batch the halo frame (slot 0) is CARRIED on the device -- pyramids, keypoints, descriptors
        # of the previous batch's last frame --: it is neither copied on the host nor sent over PCIe again
        carry = self.chunk > 0
        s0 = 1 if carry else 0
        self.ctx.upload_frames(b, self.pin[b][0][s0:self.n], self.pin[b][1][s0:self.n], first_slot=s0)
        self.uploaded[b] = True
        self.ctx.track_uploaded_async(b, self.n, continue_chain=carry, carry_frame=carry)
        self.outstanding.append(self.n - 1)
        self.chunk += 1
        nb = b ^ 1
        if self.uploaded[nb]:
            self.ctx.wait_upload(nb)       # its page-locked memory is written next
        self.buf, self.n = nb, 1
        return done

    def poll(self):
        """Records (numpy STEP_DTYPE arrays) of the micro-batches that have completed; never waits."""
        done = []
        while self.outstanding and self.ctx.results_ready() > 0:
            done.append(self._collect())
        return done

    def push(self, left, right):
        """Hands one stereo frame over (the images are copied).  Returns the records completed meanwhile."""
        self.pin[self.buf][0][self.n, :, :self.w] = left
        self.pin[self.buf][1][self.n, :, :self.w] = right
        return self.commit()

    def next_slot(self):
        """The page-locked rows the NEXT frame goes to, (left, right) views of shape (height, width): a producer that can
        write there directly (a camera driver, a decoder -- RunBatched's decoder threads do) saves push()'s copy; follow
        it with commit()."""
        return self.pin[self.buf][0][self.n, :, :self.w], self.pin[self.buf][1][self.n, :, :self.w]

    def commit(self):
        """The frame written into next_slot() is complete.  Returns the records completed meanwhile."""
        self.n += 1
        done = self._submit() if self.n == self.depth + 1 else []
        return done + self.poll()

    def flush(self):
        """Submits the partial micro-batch and waits for everything in flight."""
        done = self._submit()
        while self.outstanding:
            done.append(self._collect())
        return done


def concat_records(chunks, dtype):
    return np.concatenate(chunks) if chunks else np.zeros(0, dtype=dtype)
