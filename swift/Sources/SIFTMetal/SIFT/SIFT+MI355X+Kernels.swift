import CSiftmi

/// The reference's other public compute types, over the same C ABI: `DifferenceOfGaussians`
/// (SIFT/DifferenceOfGaussians.swift:20, 233, 346) and `SIFTDescriptorKernel` (Metal Compute/SIFTDescriptorKernel.swift:13-34).
/// In the reference they are stages a caller may drive by itself (DifferenceOfGaussiansTests.swift:15-270 encodes the pyramid alone and
/// diffs its textures); here a context computes pyramid and keypoints in one launch sequence, so the pyramid object owns a context,
/// `encode` runs it on a frame, and the textures are read back from the resident Gaussian stack (the DoG layers are never stored:
/// `siftmi_copy_dog` forms G[s + 1] - G[s], the single f32 subtraction of Metal/Subtract.metal:17-19).
public final class DifferenceOfGaussians {

    public struct Configuration {
        var inputDimensions: IntegralSize
        var sigmaMinimum: Float = 0.8                 // DifferenceOfGaussians.swift:28-46, the same defaults
        var deltaMinimum: Float = 0.5
        var sigmaInput: Float = 0.5
        var numberOfOctaves: Int = 7
        var numberOfScalesPerOctave: Int = 3
        public init(inputDimensions: IntegralSize) { self.inputDimensions = inputDimensions }
    }

    /// One octave's view (DifferenceOfGaussians.swift:55-66: o, delta, numberOfScales, sigmas, size + its two texture arrays).
    public struct Octave {
        public let o: Int
        public let delta: Float
        public let numberOfScales: Int
        public let sigmas: [Float]
        public let size: IntegralSize
        unowned let owner: DifferenceOfGaussians

        /// Gaussian layer `scale` (0 ..< numberOfScales + 3), dense rows of `size.width` floats (the .r32Float slice of gaussianTextures).
        public func gaussianTexture(scale: Int) -> [Float] {
            var out = [Float](repeating: 0, count: size.width * size.height)
            let rc = siftmi_copy_gaussian(owner.ctx, 0, Int32(o), Int32(scale), &out)
            precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
            return out
        }

        /// Difference layer `scale` (0 ..< numberOfScales + 2) = G[scale + 1] - G[scale] (the slice of differenceTextures).
        public func differenceTexture(scale: Int) -> [Float] {
            var out = [Float](repeating: 0, count: size.width * size.height)
            let rc = siftmi_copy_dog(owner.ctx, 0, Int32(o), Int32(scale), &out)
            precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
            return out
        }
    }

    let configuration: Configuration
    var ctx: OpaquePointer?
    public private(set) var octaves: [Octave] = []

    /// `device`: HIP device ordinal (the Metal build takes an MTLDevice).
    public init(device: Int32 = 0, configuration: Configuration) {
        self.configuration = configuration
        var cfg = siftmi_config()
        siftmi_default_config(&cfg, Int32(configuration.inputDimensions.width), Int32(configuration.inputDimensions.height))
        cfg.sigma_min = configuration.sigmaMinimum
        cfg.delta_min = configuration.deltaMinimum
        cfg.sigma_in = configuration.sigmaInput
        cfg.n_octaves = Int32(configuration.numberOfOctaves)
        cfg.nspo = Int32(configuration.numberOfScalesPerOctave)
        let rc = siftmi_create(&cfg, device, &ctx)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        for o in 0 ..< configuration.numberOfOctaves {
            var w: Int32 = 0, h: Int32 = 0, delta: Float = 0
            siftmi_octave_size(ctx, Int32(o), &w, &h, &delta)
            let sigmas = (0 ..< configuration.numberOfScalesPerOctave + 3).map { s -> Float in
                var v: Float = 0
                siftmi_get_sigma(ctx, Int32(o), Int32(s), &v)
                return v
            }
            octaves.append(Octave(o: o, delta: delta, numberOfScales: configuration.numberOfScalesPerOctave, sigmas: sigmas,
                                  size: IntegralSize(width: Int(w), height: Int(h)), owner: self))
        }
    }

    deinit { siftmi_destroy(ctx) }

    /// encode(commandBuffer:originalTexture:) (DifferenceOfGaussians.swift:346-355): luma, 2x bilinear, seed blur, every octave's layers.
    /// `pixels`: BGRA8, `bytesPerRow` apart.  Synchronous (the reference's caller commits and waits); the keypoints the same launch
    /// sequence finds are dropped here -- `SIFT.getKeypoints` is the entry that returns them.
    public func encode(_ pixels: UnsafeRawPointer, bytesPerRow: Int) {
        var kps: UnsafePointer<siftmi_keypoint>? = nil
        var counts = [Int32](repeating: 0, count: configuration.numberOfOctaves)
        let rc = siftmi_detect(ctx, pixels, Int32(SIFTMI_FMT_BGRA8.rawValue), bytesPerRow, 0, &kps, &counts)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
    }
}

/// `SIFTDescriptorKernel` (Metal Compute/SIFTDescriptorKernel.swift:13-34): the descriptor stage by itself, on the pyramid a
/// `DifferenceOfGaussians` holds.  The reference's encode takes the stage's buffers (parameters, gradient textures, (keypoint, theta)
/// inputs, 524-byte results); here orientation and descriptor are one call on the resident Gaussian stack (gradients are formed on
/// demand, never stored), so the inputs are the keypoints themselves and the results come back as `SIFTDescriptor`s.
public final class SIFTDescriptorKernel {

    public init(device: Int32 = 0) {}

    public func encode(pyramid: DifferenceOfGaussians, keypointOctaves: [[SIFTKeypoint]]) -> [[SIFTDescriptor]] {
        let n = pyramid.octaves.count
        precondition(keypointOctaves.count == n)
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
        var dcounts = [Int32](repeating: 0, count: n)
        let rc = siftmi_describe(pyramid.ctx, flat, counts, &out, &dcounts)
        precondition(rc == SIFTMI_OK.rawValue, String(cString: siftmi_last_error()))
        var result = [[SIFTDescriptor]](), p = 0
        for o in 0 ..< n {
            result.append((0 ..< Int(dcounts[o])).map { i in
                var d = out![p + i]
                let features = withUnsafeBytes(of: &d.features) { $0.map { Int($0) } }
                return SIFTDescriptor(keypoint: keypointOctaves[o][Int(d.keypoint)], theta: d.theta, features: IntVector(features))
            })
            p += Int(dcounts[o])
        }
        return result
    }
}
