import CSiftmi

public final class SIFT {

    public struct Configuration {
        var inputSize: IntegralSize
        public init(inputSize: IntegralSize) { self.inputSize = inputSize }
    }

    var ctx: OpaquePointer?                                    // internal: SIFTStream (SIFT+MI355X+Stream.swift) borrows it
    let octaveCount: Int

    /// `device`: HIP device ordinal (the Metal build takes an MTLDevice here); `batch`: frames processed in
    /// lock-step per launch when the object feeds a `SIFTStream` (1 for getKeypoints / getDescriptors).
    public init(device: Int32 = 0, configuration: Configuration, batch: Int = 1) {
        var cfg = siftmi_config()
        siftmi_default_config(&cfg, Int32(configuration.inputSize.width), Int32(configuration.inputSize.height))
        cfg.max_batch = Int32(batch)
        octaveCount = Int(cfg.n_octaves)                       // 7, as DifferenceOfGaussians.swift:41
        let rc = siftmi_create(&cfg, device, &ctx)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
    }

    deinit { siftmi_destroy(ctx) }

    /// `pixels`: BGRA8, `bytesPerRow` apart (what `MTLTexture.getBytes` / a CVPixelBuffer gives).
    public func getKeypoints(_ pixels: UnsafeRawPointer, bytesPerRow: Int) -> [[SIFTKeypoint]] {
        var out: UnsafePointer<siftmi_keypoint>? = nil
        var counts = [Int32](repeating: 0, count: octaveCount)
        let rc = siftmi_detect(ctx, pixels, Int32(SIFTMI_FMT_BGRA8.rawValue), bytesPerRow, 0, &out, &counts)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        var result = [[SIFTKeypoint]](), p = 0
        for o in 0 ..< octaveCount {
            result.append((0 ..< Int(counts[o])).map { i in
                let k = out![p + i]
                return SIFTKeypoint(octave: Int(k.octave), scale: Int(k.scale), subScale: k.sub_scale,
                                    scaledCoordinate: SIMD2<Int>(Int(k.x), Int(k.y)),
                                    absoluteCoordinate: SIMD2<Float>(k.abs_x, k.abs_y),
                                    normalizedCoordinate: SIMD2<Float>(k.norm_x, k.norm_y),
                                    sigma: k.sigma, value: k.value)
            })
            p += Int(counts[o])
        }
        return result
    }

    public func getDescriptors(keypointOctaves: [[SIFTKeypoint]]) -> [[SIFTDescriptor]] {
        precondition(keypointOctaves.count == octaveCount)            // SIFT.swift:208
        var flat = [siftmi_keypoint]()
        var counts = [Int32]()
        for octave in keypointOctaves {
            counts.append(Int32(octave.count))
            for k in octave {
                flat.append(siftmi_keypoint(octave: Int32(k.octave), scale: Int32(k.scale), sub_scale: k.subScale,
                                            x: Int32(k.scaledCoordinate.x), y: Int32(k.scaledCoordinate.y),
                                            abs_x: k.absoluteCoordinate.x, abs_y: k.absoluteCoordinate.y,
                                            norm_x: k.normalizedCoordinate.x, norm_y: k.normalizedCoordinate.y,
                                            sigma: k.sigma, value: k.value))
            }
        }
        var out: UnsafePointer<siftmi_descriptor>? = nil
        var dcounts = [Int32](repeating: 0, count: octaveCount)
        let rc = siftmi_describe(ctx, flat, counts, &out, &dcounts)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        var result = [[SIFTDescriptor]](), p = 0
        for o in 0 ..< octaveCount {
            result.append((0 ..< Int(dcounts[o])).map { i in
                var d = out![p + i]
                let features = withUnsafeBytes(of: &d.features) { $0.map { Int($0) } }   // 128 x 0...255
                return SIFTDescriptor(keypoint: keypointOctaves[o][Int(d.keypoint)], theta: d.theta,
                                      features: IntVector(features))                     // SIFTDescriptor.swift:26-34
            })
            p += Int(dcounts[o])
        }
        return result
    }
}
