import CSiftmi

/// A stream of frame batches through one `SIFT` object with several batches in flight, and the exchange of
/// results between the GPUs of a node.  The Metal build has no counterpart: its `SIFT` drives one command
/// queue and blocks after every stage (SIFT.swift:139-175).
public final class SIFTStream {

    public struct Step {
        public let number: Int64
        public let keypoints: UnsafeBufferPointer<siftmi_keypoint>      // packed, ordered (frame, octave)
        public let descriptors: UnsafeBufferPointer<siftmi_descriptor>
        public let counts: UnsafeBufferPointer<Int32>                   // [2][framesPerStep][octaves]
        public let truncated: Bool
    }

    private var stream: OpaquePointer?
    private var exchange: OpaquePointer?
    private let sift: SIFT                                              // keeps the borrowed context alive
    public let framesPerStep: Int

    /// `sift` must have been created with `max_batch` frames in lock-step (SIFT.init(device:configuration:batch:)).
    public init(sift: SIFT, framesPerStep: Int, stepsInFlight: Int = 2) {
        self.sift = sift
        self.framesPerStep = framesPerStep
        var scfg = siftmi_stream_config()
        siftmi_stream_default_config(&scfg, Int32(framesPerStep))
        scfg.steps_in_flight = Int32(stepsInFlight)
        let rc = siftmi_stream_create(sift.ctx, &scfg, &stream)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
    }

    deinit {
        if exchange != nil { siftmi_exchange_destroy(exchange) }
        siftmi_stream_destroy(stream)
    }

    /// `pixels`: `framesPerStep` BGRA8 frames in page-locked memory (`siftmi_host_alloc`); returns the step number.
    /// The buffer may be refilled after `waitUpload(step)`.
    @discardableResult
    public func submit(_ pixels: UnsafeRawPointer, bytesPerRow: Int, bytesPerFrame: Int) -> Int64 {
        var step: Int64 = -1
        let rc = siftmi_stream_submit_host(stream, pixels, bytesPerRow, bytesPerFrame, &step)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        return step
    }

    public func waitUpload(_ step: Int64) {
        precondition(siftmi_stream_wait_upload(stream, step) == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
    }

    /// Results of the step `back` steps before the last submitted one, in page-locked host memory owned by the
    /// stream (valid until `result_sets` further steps have been submitted).  Blocks until that step is done.
    public func result(back: Int = 0, octaves: Int) -> Step {
        var r = siftmi_step_host()
        let rc = siftmi_stream_result_host(stream, Int32(back), &r)
        precondition(rc == SIFTMI_OK.rawValue || rc == SIFTMI_E_CAPACITY.rawValue, String(cString: siftmi_last_error()))
        return Step(number: r.step,
                    keypoints: UnsafeBufferPointer(start: r.keypoints, count: Int(r.n_keypoints)),
                    descriptors: UnsafeBufferPointer(start: r.descriptors, count: Int(r.n_descriptors)),
                    counts: UnsafeBufferPointer(start: r.counts, count: 2 * framesPerStep * octaves),
                    truncated: r.overflow_flags != 0)
    }

    // MARK: exchange between the GPUs of a node (one process per GPU, RCCL over xGMI)

    /// A collective that cannot complete: a rank died or hangs (every host wait of the exchange is bounded:
    /// `SIFTMI_EXCHANGE_TIMEOUT_S`, `setExchangeTimeout`), the communicator reported an error, or the call was out of order.
    /// After a time-out the communicator is aborted and every later exchange call throws the same message; results of the
    /// local stream (`result(back:octaves:)`) stay available.
    public struct ExchangeError: Error, CustomStringConvertible {
        public let code: Int32
        public let description: String
    }

    private func check(_ rc: Int32) throws {
        if rc != SIFTMI_OK.rawValue { throw ExchangeError(code: rc, description: String(cString: siftmi_last_error())) }
    }

    /// Rank 0 creates the id and hands it to the other ranks out of band.
    public static func makeExchangeID() throws -> [UInt8] {
        var id = [UInt8](repeating: 0, count: Int(SIFTMI_UNIQUE_ID_BYTES))
        let rc = siftmi_exchange_unique_id(&id)
        if rc != SIFTMI_OK.rawValue { throw ExchangeError(code: rc, description: String(cString: siftmi_last_error())) }
        return id
    }

    /// Collective over all `world` ranks.  Throws when the communicator does not report `world` ranks and this rank number.
    public func joinExchange(id: [UInt8], rank: Int, world: Int) throws {
        try check(siftmi_exchange_create(stream, id, Int32(rank), Int32(world), &exchange))
    }

    /// (ranks, rank) as the communicator itself reports them (ncclCommCount / ncclCommUserRank).
    public func exchangeRanks() throws -> (ranks: Int, rank: Int) {
        var n: Int32 = 0, r: Int32 = -1
        try check(siftmi_exchange_ranks(exchange, &n, &r))
        return (Int(n), Int(r))
    }

    /// Deadline of every host wait of the exchange, in seconds (default 120, or `SIFTMI_EXCHANGE_TIMEOUT_S`).
    public func setExchangeTimeout(seconds: Double) throws {
        try check(siftmi_exchange_set_timeout(exchange, seconds))
    }

    /// Collective: all-gather the last submitted step's keypoints and descriptors (asynchronous, side stream).
    public func gather() throws {
        try check(siftmi_exchange_gather(exchange, 0))
    }

    /// Bounded wait for every gather enqueued so far (before a process-wide barrier or a device synchronisation,
    /// which would otherwise wait for ever on a collective a dead rank never joins).
    public func waitForExchange() throws {
        try check(siftmi_exchange_wait(exchange))
    }

    /// Device view of every rank's results of the gather `back` gathers ago; `complete == 0` until the next
    /// `gather()` / `finishExchange()` when some rank held more records than were sent.
    public func gathered(back: Int = 0, wait: Bool = true) throws -> siftmi_gathered {
        var g = siftmi_gathered()
        let none = UnsafeMutableRawPointer(bitPattern: -1)              // SIFTMI_NO_STREAM
        try check(siftmi_exchange_result(exchange, Int32(back), &g, none, wait ? 1 : 0))
        return g
    }

    /// Collective, end of stream.
    public func finishExchange() throws {
        try check(siftmi_exchange_finish(exchange, nil, nil))
    }
}
